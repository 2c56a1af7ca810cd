// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see spiel_oracle.h).
// MCTS, CFR / CFR+, external-sampling MCCFR and the exploitability judge.
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

#include "spiel_oracle.h"

namespace osg_oracle {

// ============================================================================
// Policies
// ============================================================================
ActionsAndProbs UniformPolicy::GetStatePolicy(const State& state,
                                              Player p) const {
  // policy.cc UniformStatePolicy: equal mass on the legal actions of `p`.
  std::vector<Action> legal = state.LegalActions(p);
  ActionsAndProbs out;
  for (Action a : legal) out.push_back({a, 1.0 / legal.size()});
  return out;
}

namespace {
void WalkInfostates(const State& s,
                    const std::function<void(const State&)>& visit) {
  if (s.IsTerminal()) return;
  if (s.IsChanceNode()) {
    for (const auto& ap : s.ChanceOutcomes()) WalkInfostates(*s.Child(ap.first), visit);
    return;
  }
  visit(s);
  for (Action a : s.LegalActions()) WalkInfostates(*s.Child(a), visit);
}
}  // namespace

TabularPolicy GetUniformPolicy(const Game& game) {
  TabularPolicy pol;
  WalkInfostates(*game.NewInitialState(), [&](const State& s) {
    std::vector<Action> legal = s.LegalActions();
    ActionsAndProbs ap;
    for (Action a : legal) ap.push_back({a, 1.0 / legal.size()});
    pol.Table()[s.InformationStateString()] = ap;
  });
  return pol;
}
TabularPolicy GetFirstActionPolicy(const Game& game) {
  TabularPolicy pol;
  WalkInfostates(*game.NewInitialState(), [&](const State& s) {
    std::vector<Action> legal = s.LegalActions();
    ActionsAndProbs ap;
    for (size_t i = 0; i < legal.size(); ++i) ap.push_back({legal[i], i == 0 ? 1.0 : 0.0});
    pol.Table()[s.InformationStateString()] = ap;
  });
  return pol;
}
TabularPolicy KuhnOptimalPolicy(double alpha) {  // kuhn_poker.cc:451-474
  ORACLE_CHECK(alpha >= 0 && alpha <= 1.0 / 3);
  std::unordered_map<std::string, ActionsAndProbs> t;
  t["0"] = {{0, 1 - alpha}, {1, alpha}};
  t["0pb"] = {{0, 1}, {1, 0}};
  t["1"] = {{0, 1}, {1, 0}};
  t["1pb"] = {{0, 2. / 3. - alpha}, {1, 1. / 3. + alpha}};
  t["2"] = {{0, 1 - 3 * alpha}, {1, 3 * alpha}};
  t["2pb"] = {{0, 0}, {1, 1}};
  t["0p"] = {{0, 2. / 3.}, {1, 1. / 3.}};
  t["0b"] = {{0, 1}, {1, 0}};
  t["1p"] = {{0, 1}, {1, 0}};
  t["1b"] = {{0, 2. / 3.}, {1, 1. / 3.}};
  t["2p"] = {{0, 0}, {1, 1}};
  t["2b"] = {{0, 0}, {1, 1}};
  return TabularPolicy(t);
}

// ============================================================================
// MCTS
// ============================================================================
std::vector<double> RandomRolloutEvaluator::Evaluate(const State& state) {
  // mcts.cc:43-72.  absl::Uniform(rng, 0u, n) is replaced by a multiply-shift
  // draw from the same mt19937 (the abseil stream is unpinned, see header).
  std::vector<double> total;
  for (int i = 0; i < n_rollouts_; ++i) {
    std::unique_ptr<State> w = state.Clone();
    while (!w->IsTerminal()) {
      if (w->IsChanceNode()) {
        double z = (rng_() >> 5) * (1.0 / 134217728.0);  // 27 bits in [0,1)
        w->ApplyAction(SampleAction(w->ChanceOutcomes(), z).first);
      } else {
        std::vector<Action> legal = w->LegalActions();
        uint64_t pick = (static_cast<uint64_t>(rng_()) * legal.size()) >> 32;
        w->ApplyAction(legal[pick]);
      }
    }
    std::vector<double> r = w->Returns();
    if (total.empty()) {
      total.swap(r);
    } else {
      for (size_t k = 0; k < total.size(); ++k) total[k] += r[k];
    }
  }
  for (double& v : total) v /= n_rollouts_;
  return total;
}

ActionsAndProbs RandomRolloutEvaluator::Prior(const State& state) {
  if (state.IsChanceNode()) return state.ChanceOutcomes();  // mcts.cc:74-87
  std::vector<Action> legal = state.LegalActions();
  ActionsAndProbs prior;
  prior.reserve(legal.size());
  for (Action a : legal) prior.emplace_back(a, 1.0 / legal.size());
  return prior;
}

double SearchNode::UCTValue(int parent_n, double c) const {  // mcts.cc:90-101
  if (!outcome.empty()) return outcome[player];
  if (explore_count == 0) return std::numeric_limits<double>::infinity();
  return total_reward / explore_count +
         c * std::sqrt(std::log(parent_n) / explore_count);
}
double SearchNode::PUCTValue(int parent_n, double c) const {  // mcts.cc:103-112
  if (!outcome.empty()) return outcome[player];
  return (explore_count != 0 ? total_reward / explore_count : 0) +
         c * prior * std::sqrt(parent_n) / (explore_count + 1);
}
bool SearchNode::CompareFinal(const SearchNode& b) const {  // mcts.cc:114-125
  double mine = (player >= 0 && player < static_cast<int>(outcome.size()))
                    ? outcome[player] : 0;
  double theirs = (b.player >= 0 && b.player < static_cast<int>(b.outcome.size()))
                      ? b.outcome[b.player] : 0;
  if (mine != theirs) return mine < theirs;
  if (explore_count != b.explore_count) return explore_count < b.explore_count;
  return total_reward < b.total_reward;
}
const SearchNode& SearchNode::BestChild() const {  // mcts.cc:127-143
  return *std::max_element(
      children.begin(), children.end(),
      [](const SearchNode& a, const SearchNode& b) { return a.CompareFinal(b); });
}

MCTSBot::MCTSBot(const Game& game, std::shared_ptr<Evaluator> evaluator,
                 double uct_c, int max_simulations, int64_t max_memory_mb,
                 bool solve, int seed, bool /*verbose*/,
                 ChildSelectionPolicy policy, bool dont_return_chance_node)
    : uct_c_(uct_c),
      max_simulations_(max_simulations),
      // max_memory_mb < 0: test hook of the restatement — max_nodes_ given directly as -max_memory_mb, so that a
      // device search (whose nodes are 24 bytes, not sizeof(SearchNode)) can be replayed at the same node budget
      max_nodes_(max_memory_mb < 0 ? static_cast<int>(-max_memory_mb)
                                   : static_cast<int>((max_memory_mb << 20) / sizeof(SearchNode) + 1)),
      gc_limit_(5),
      solve_(solve),
      max_utility_(game.MaxUtility()),
      dont_return_chance_node_(dont_return_chance_node),
      rng_(seed),
      child_selection_policy_(policy),
      evaluator_(std::move(evaluator)) {}

Action MCTSBot::Step(const State& state) {  // mcts.cc:233-266
  std::unique_ptr<SearchNode> root = MCTSearch(state);
  if (max_simulations_ <= 1) {
    ActionsAndProbs prior = evaluator_->Prior(state);
    double z = (rng_() >> 5) * (1.0 / 134217728.0);
    return SampleAction(prior, z).first;
  }
  return root->BestChild().action;
}

std::unique_ptr<State> MCTSBot::ApplyTreePolicy(
    SearchNode* root, const State& state, std::vector<SearchNode*>* path) {
  // mcts.cc:273-351
  path->push_back(root);
  std::unique_ptr<State> w = state.Clone();
  SearchNode* node = root;
  uint64_t ph = PathHashRoot();
  while ((!w->IsTerminal() && node->explore_count > 0) ||
         (w->IsChanceNode() && dont_return_chance_node_)) {
    if (node->children.empty()) {
      ActionsAndProbs legal = evaluator_->Prior(*w);
      if (counter_ && c_layout_ == 2) {  // order by key (the wave kernel's tie-break order)
        const uint64_t ob = OrderBase(c_seed_, c_root_);
        std::sort(legal.begin(), legal.end(), [&](const std::pair<Action, double>& a,
                                                  const std::pair<Action, double>& b) {
          // (key, action): the key's low byte is the action's low byte, so only actions 256 apart — the boards above
          // 256 cells — can tie, and then the smaller action goes first (the kernel's children are in action order)
          const uint32_t ka = OrderKey(ob, ph, static_cast<int>(a.first)), kb = OrderKey(ob, ph, static_cast<int>(b.first));
          return ka != kb ? ka < kb : a.first < b.first;
        });
      } else if (counter_) {  // Fisher-Yates on the counter stream (the device's shuffle)
        for (int i = static_cast<int>(legal.size()) - 1; i >= 1; --i)
          std::swap(legal[i], legal[trng_.Below(static_cast<uint32_t>(i + 1))]);
      } else {
        std::shuffle(legal.begin(), legal.end(), rng_);
      }
      Player mover = w->CurrentPlayer();
      node->children.reserve(legal.size());
      for (const auto& ap : legal)
        node->children.emplace_back(ap.first, mover, ap.second);
      nodes_ += static_cast<int>(node->children.capacity());
    }
    SearchNode* chosen = nullptr;
    if (w->IsChanceNode()) {
      double z = counter_ ? trng_.Unit() : (rng_() >> 5) * (1.0 / 134217728.0);
      Action a = SampleAction(w->ChanceOutcomes(), z).first;
      for (SearchNode& c : node->children)
        if (c.action == a) {
          chosen = &c;
          break;
        }
    } else {
      double best = -std::numeric_limits<double>::infinity();
      for (SearchNode& c : node->children) {
        double v = child_selection_policy_ == ChildSelectionPolicy::UCT
                       ? c.UCTValue(node->explore_count, uct_c_)
                       : c.PUCTValue(node->explore_count, uct_c_);
        if (v > best) {  // strict: first maximum wins
          best = v;
          chosen = &c;
        }
      }
    }
    ORACLE_CHECK(chosen != nullptr);
    w->ApplyAction(chosen->action);
    ph = PathHashChild(ph, static_cast<int>(chosen->action));
    node = chosen;
    path->push_back(node);
  }
  return w;
}

std::unique_ptr<SearchNode> MCTSBot::MCTSearch(const State& state) {
  // mcts.cc:353-467
  nodes_ = 1;
  gc_limit_ = 5;
  auto root = std::make_unique<SearchNode>(kInvalidAction, state.CurrentPlayer(), 1);
  std::vector<SearchNode*> path;
  path.reserve(64);
  for (int sim = 0; sim < max_simulations_; ++sim) {
    path.clear();
    if (counter_) trng_ = CounterRng(c_seed_ ^ kTreeSalt, c_root_, static_cast<uint64_t>(sim));
    std::unique_ptr<State> leaf = ApplyTreePolicy(root.get(), state, &path);
    std::vector<double> returns;
    bool solved;
    if (leaf->IsTerminal()) {
      returns = leaf->Returns();
      path.back()->outcome = returns;
      solved = solve_;
    } else {
      returns = (counter_ && !c_use_evaluator_) ? CounterEvaluate(*leaf, sim) : evaluator_->Evaluate(*leaf);
      solved = false;
    }
    while (!path.empty()) {  // backup, mcts.cc:383-435
      int idx = static_cast<int>(path.size()) - 1;
      SearchNode* node = path[idx];
      while (path[idx]->player == kChancePlayerId) --idx;
      node->total_reward += returns[path[idx]->player];
      node->explore_count += 1;
      path.pop_back();
      if (solved && !node->children.empty()) {
        Player mover = node->children[0].player;
        if (mover == kChancePlayerId) {
          const std::vector<double>& first = node->children[0].outcome;
          bool same = !first.empty();
          for (size_t i = 1; same && i < node->children.size(); ++i)
            same = node->children[i].outcome == first;
          if (same) node->outcome = first;
          else solved = false;
        } else {
          const SearchNode* best = nullptr;
          bool all_solved = true;
          for (const SearchNode& c : node->children) {
            if (c.outcome.empty()) all_solved = false;
            else if (best == nullptr || c.outcome[mover] > best->outcome[mover])
              best = &c;
          }
          if (best != nullptr && (all_solved || best->outcome[mover] == max_utility_))
            node->outcome = best->outcome;
          else
            solved = false;
        }
      }
    }
    if (!root->outcome.empty() || root->children.size() == 1) break;
    if (max_nodes_ > 1 && nodes_ >= max_nodes_) {  // mcts.cc:441-463
      GarbageCollect(root.get());
      gc_limit_ = static_cast<int>(gc_limit_ * (nodes_ > max_nodes_ / 2 ? 1.25 : 0.9));
      gc_limit_ = std::max(5, gc_limit_);
    }
  }
  return root;
}

std::vector<double> MCTSBot::CounterEvaluate(const State& state, int sim) const {
  // RandomRolloutEvaluator::Evaluate (mcts.cc:43-72) on the device's counter streams.
  std::vector<double> total(state.NumPlayers(), 0.0);
  const Game& game = *state.GetGame();
  const bool keyed_fill = c_layout_ == 2 && game.ShortName() == "hex" &&
                          game.NumDistinctActions() == game.ObservationTensorShape()[1] * game.ObservationTensorShape()[2];
  for (int r = 0; r < c_rollouts_; ++r) {
    CounterRng rng(c_seed_, c_root_, static_cast<uint64_t>(sim) * c_rollouts_ + r);
    std::unique_ptr<State> w = state.Clone();
    if (keyed_fill) {
      // Order the empty cells by key; the player to move plays the ceil(m/2) smallest in
      // ascending order on plies 0, 2, 4, ..., the opponent the others on plies 1, 3, 5, ...
      // (a fixed re-indexing of a uniformly random order: again uniformly random), with the
      // reference's own loop — stop at the first terminal state (mcts.cc:45-56).
      const uint64_t fb = FillBase(c_seed_, c_root_, static_cast<uint64_t>(sim) * c_rollouts_ + r);
      std::vector<Action> cells = w->LegalActions();
      std::sort(cells.begin(), cells.end(), [&](Action a, Action b) {
        const uint64_t ka = FillKey(fb, static_cast<int>(a)), kb = FillKey(fb, static_cast<int>(b));
        return ka != kb ? ka < kb : a < b;   // (FillKey, cell): cells 256 apart may share a key on the largest boards
      });
      const size_t h = (cells.size() + 1) / 2;
      for (size_t t = 0; !w->IsTerminal(); ++t) {
        ORACLE_CHECK(t < cells.size());
        w->ApplyAction(t % 2 == 0 ? cells[t / 2] : cells[h + t / 2]);
      }
    }
    while (!w->IsTerminal()) {
      if (w->IsChanceNode()) {
        w->ApplyAction(SampleAction(w->ChanceOutcomes(), rng.Unit()).first);
      } else {
        std::vector<Action> legal = w->LegalActions();
        w->ApplyAction(legal[rng.Below(static_cast<uint32_t>(legal.size()))]);
      }
    }
    std::vector<double> ret = w->Returns();
    for (size_t k = 0; k < total.size(); ++k) total[k] += ret[k];
  }
  for (double& v : total) v /= c_rollouts_;
  return total;
}

void MCTSBot::GarbageCollect(SearchNode* node) {  // mcts.cc:469-482
  if (node->children.empty()) return;
  bool clear = node->explore_count < gc_limit_;
  for (SearchNode& c : node->children) GarbageCollect(&c);
  if (clear) {
    nodes_ -= static_cast<int>(node->children.capacity());
    node->children.clear();
    node->children.shrink_to_fit();
  }
}

// ============================================================================
// CFR
// ============================================================================
void CFRInfoStateValues::ApplyRegretMatching() {  // cfr.cc:596-615
  double positive = 0.0;
  for (int a = 0; a < num_actions(); ++a)
    if (cumulative_regrets[a] > 0) positive += cumulative_regrets[a];
  for (int a = 0; a < num_actions(); ++a) {
    if (positive > 0) {
      current_policy[a] = cumulative_regrets[a] > 0 ? cumulative_regrets[a] / positive : 0;
    } else {
      current_policy[a] = 1.0 / legal_actions.size();
    }
  }
}
int CFRInfoStateValues::SampleActionIndex(double epsilon, double z) {
  double acc = 0;  // cfr.cc:617-628
  for (int a = 0; a < static_cast<int>(current_policy.size()); ++a) {
    double p = epsilon * 1.0 / current_policy.size() + (1.0 - epsilon) * current_policy[a];
    if (z >= acc && z < acc + p) return a;
    acc += p;
  }
  Fatal("SampleActionIndex: sum of probs is " + std::to_string(acc));
}

ActionsAndProbs CFRAveragePolicy::FromValues(const CFRInfoStateValues& v) {
  // cfr.cc:104-125
  ActionsAndProbs out;
  double total = 0.0;
  for (int a = 0; a < v.num_actions(); ++a) total += v.cumulative_policy[a];
  if (total == 0.0) {
    for (Action a : v.legal_actions) out.push_back({a, 1. / v.num_actions()});
    return out;
  }
  for (int a = 0; a < v.num_actions(); ++a)
    out.push_back({v.legal_actions[a], v.cumulative_policy[a] / total});
  return out;
}
ActionsAndProbs CFRAveragePolicy::GetStatePolicy(const State& state, Player p) const {
  auto it = info_states_.find(state.InformationStateString(p));
  if (it == info_states_.end()) {
    if (default_policy_) return default_policy_->GetStatePolicy(state, p);
    Fatal("No policy found, and no default policy.");
  }
  return FromValues(it->second);
}
ActionsAndProbs CFRAveragePolicy::GetStatePolicy(const std::string& key) const {
  auto it = info_states_.find(key);
  if (it == info_states_.end()) {
    if (default_policy_) return default_policy_->GetStatePolicy(key);
    Fatal("No policy found, and no default policy.");
  }
  return FromValues(it->second);
}
TabularPolicy CFRAveragePolicy::AsTabular() const {
  TabularPolicy pol;
  for (const auto& kv : info_states_) pol.Table()[kv.first] = FromValues(kv.second);
  return pol;
}
ActionsAndProbs CFRCurrentPolicy::GetStatePolicy(const std::string& key) const {
  auto it = info_states_.find(key);
  if (it == info_states_.end()) Fatal("No policy found, and no default policy.");
  ActionsAndProbs out;
  for (int a = 0; a < it->second.num_actions(); ++a)
    out.push_back({it->second.legal_actions[a], it->second.current_policy[a]});
  return out;
}

CFRSolverBase::CFRSolverBase(const Game& game, bool alternating_updates,
                             bool linear_averaging, bool regret_matching_plus)
    : game_(game.shared_from_this()),
      root_state_(game.NewInitialState()),
      root_reach_probs_(game.NumPlayers() + 1, 1.0),
      regret_matching_plus_(regret_matching_plus),
      alternating_updates_(alternating_updates),
      linear_averaging_(linear_averaging),
      chance_player_(game.NumPlayers()) {
  InitializeInfostateNodes(*root_state_);  // cfr.cc:209
}

void CFRSolverBase::InitializeInfostateNodes(const State& s) {  // cfr.cc:234-261
  if (s.IsTerminal()) return;
  if (s.IsChanceNode()) {
    for (const auto& ap : s.ChanceOutcomes()) InitializeInfostateNodes(*s.Child(ap.first));
    return;
  }
  std::string key = s.InformationStateString(s.CurrentPlayer());
  std::vector<Action> legal = s.LegalActions();
  info_states_[key] = CFRInfoStateValues(legal);
  for (Action a : legal) InitializeInfostateNodes(*s.Child(a));
}

void CFRSolverBase::EvaluateAndUpdatePolicy() {  // cfr.cc:263-282
  ++iteration_;
  if (alternating_updates_) {
    for (Player p = 0; p < game_->NumPlayers(); ++p) {
      ComputeCounterFactualRegret(*root_state_, p, root_reach_probs_);
      if (regret_matching_plus_) ApplyRegretMatchingPlusReset();
      ApplyRegretMatching();
    }
  } else {
    ComputeCounterFactualRegret(*root_state_, -1, root_reach_probs_);
    if (regret_matching_plus_) ApplyRegretMatchingPlusReset();
    ApplyRegretMatching();
  }
}

std::vector<double> CFRSolverBase::ComputeCounterFactualRegret(
    const State& state, int alternating_player, const std::vector<double>& reach) {
  // cfr.cc:331-408
  if (state.IsTerminal()) return state.Returns();
  if (state.IsChanceNode()) {
    ActionsAndProbs ap = state.ChanceOutcomes();
    std::vector<double> dist(ap.size());
    std::vector<Action> outcomes(ap.size());
    for (size_t i = 0; i < ap.size(); ++i) {
      outcomes[i] = ap[i].first;
      dist[i] = ap[i].second;
    }
    return ForActionProbs(state, alternating_player, reach, chance_player_, dist,
                          outcomes, nullptr);
  }
  bool all_zero = true;  // cfr.cc:350-355,471-479
  for (int p = 0; p < game_->NumPlayers(); ++p)
    if (reach[p] != 0.0) all_zero = false;
  if (all_zero) return std::vector<double>(game_->NumPlayers(), 0.0);

  int cur = state.CurrentPlayer();
  std::string key = state.InformationStateString();
  std::vector<Action> legal = state.LegalActions(cur);
  auto entry = info_states_.find(key);  // GetPolicy, cfr.cc:481-493
  if (entry == info_states_.end()) {
    info_states_[key] = CFRInfoStateValues(legal);
    entry = info_states_.find(key);
  }
  std::vector<double> policy = entry->second.current_policy;
  if (overrides_ && (*overrides_)[cur]) {  // GetInfoStatePolicyFromPolicy (cfr.cc:365-372,411-430)
    const Action br = (*overrides_)[cur]->at(key);
    for (size_t a = 0; a < legal.size(); ++a) policy[a] = legal[a] == br ? 1.0 : 0.0;
  }

  std::vector<double> child_utils;
  child_utils.reserve(legal.size());
  const std::vector<double> value =
      ForActionProbs(state, alternating_player, reach, cur, policy, legal, &child_utils);

  if (alternating_player < 0 || alternating_player == cur) {
    CFRInfoStateValues vals = info_states_[key];  // copy in (cfr.cc:380)
    ORACLE_CHECK(!vals.empty());
    const double self_reach = reach[cur];
    double cf_reach = 1.0;  // CounterFactualReachProb, cfr.cc:309-318
    for (int i = 0; i < static_cast<int>(reach.size()); ++i)
      if (i != cur) cf_reach *= reach[i];
    for (size_t a = 0; a < legal.size(); ++a) {
      double regret = cf_reach * (child_utils[a] - value[cur]);
      vals.cumulative_regrets[a] += regret;
      if (linear_averaging_) {
        vals.cumulative_policy[a] += iteration_ * self_reach * policy[a];
      } else {
        vals.cumulative_policy[a] += self_reach * policy[a];
      }
    }
    info_states_[key] = vals;  // write back (cfr.cc:404)
  }
  return value;
}

std::vector<double> CFRSolverBase::ForActionProbs(
    const State& state, int alternating_player, const std::vector<double>& reach,
    int current_player, const std::vector<double>& probs,
    const std::vector<Action>& actions, std::vector<double>* child_values_out) {
  // cfr.cc:443-469
  std::vector<double> value(game_->NumPlayers());
  for (size_t i = 0; i < actions.size(); ++i) {
    const double prob = probs[i];
    std::unique_ptr<State> child = state.Child(actions[i]);
    std::vector<double> child_reach(reach);
    child_reach[current_player] *= prob;
    std::vector<double> cv = ComputeCounterFactualRegret(*child, alternating_player, child_reach);
    for (size_t p = 0; p < value.size(); ++p) value[p] += prob * cv[p];
    if (child_values_out != nullptr) child_values_out->push_back(cv[current_player]);
  }
  return value;
}

void CFRSolverBase::ApplyRegretMatchingPlusReset() {  // cfr.cc:683-691
  for (auto& kv : info_states_)
    for (double& r : kv.second.cumulative_regrets)
      if (r < 0) r = 0;
}
void CFRSolverBase::ApplyRegretMatching() {  // cfr.cc:693-697
  for (auto& kv : info_states_) kv.second.ApplyRegretMatching();
}

// ============================================================================
// External-sampling MCCFR
// ============================================================================
ExternalSamplingMCCFRSolver::ExternalSamplingMCCFRSolver(const Game& game, int seed,
                                                         AverageType avg)
    : game_(game.shared_from_this()),
      rng_(seed),
      avg_type_(avg),
      dist_(0.0, 1.0),
      default_policy_(std::make_shared<UniformPolicy>()) {}

void ExternalSamplingMCCFRSolver::RunIteration() {  // external_sampling_mccfr.cc:71-80
  auto next_z = [this]() { return dist_(rng_); };
  for (Player p = 0; p < game_->NumPlayers(); ++p)
    UpdateRegretsWith(*game_->NewInitialState(), p, next_z);
  if (avg_type_ == AverageType::kFull) {
    std::vector<double> reach(game_->NumPlayers(), 1.0);
    FullUpdateAverage(*game_->NewInitialState(), reach);
  }
}

double ExternalSamplingMCCFRSolver::UpdateRegretsWith(
    const State& state, Player player, const std::function<double()>& next_z,
    const std::function<void(int, int, int)>* on_branch, int depth, int b1) {
  // external_sampling_mccfr.cc:122-186
  if (state.IsTerminal()) return state.PlayerReturn(player);
  if (state.IsChanceNode()) {
    Action a = SampleAction(state.ChanceOutcomes(), next_z()).first;
    return UpdateRegretsWith(*state.Child(a), player, next_z, on_branch, depth, b1);
  }
  Player cur = state.CurrentPlayer();
  std::string key = state.InformationStateString(cur);
  std::vector<Action> legal = state.LegalActions();
  auto ins = info_states_.insert({key, CFRInfoStateValues(legal, kInitialTableValues)});
  CFRInfoStateValues copy = ins.first->second;
  copy.ApplyRegretMatching();

  double value = 0;
  std::vector<double> child_values(legal.size(), 0);
  if (cur != player) {
    int a = copy.SampleActionIndex(0.0, next_z());
    value = UpdateRegretsWith(*state.Child(legal[a]), player, next_z, on_branch, depth, b1);
  } else {
    for (size_t a = 0; a < legal.size(); ++a) {
      const int ai = static_cast<int>(a);
      if (on_branch && depth == 0) (*on_branch)(1, ai, 0);
      if (on_branch && depth == 1) (*on_branch)(2, b1, ai);
      child_values[a] = UpdateRegretsWith(*state.Child(legal[a]), player, next_z, on_branch, depth + 1, depth == 0 ? ai : b1);
      value += copy.current_policy[a] * child_values[a];
    }
  }
  CFRInfoStateValues& row = info_states_[key];
  if (cur == player) {
    for (size_t a = 0; a < legal.size(); ++a)
      row.cumulative_regrets[a] += (child_values[a] - value);
  }
  if (avg_type_ == AverageType::kSimple && cur == ((player + 1) % game_->NumPlayers())) {
    for (size_t a = 0; a < legal.size(); ++a)
      row.cumulative_policy[a] += copy.current_policy[a];
  }
  return value;
}

void ExternalSamplingMCCFRSolver::FullUpdateAverage(const State& state,
                                                    const std::vector<double>& reach) {
  // external_sampling_mccfr.cc:188-231
  if (state.IsTerminal()) return;
  if (state.IsChanceNode()) {
    for (Action a : state.LegalActions()) FullUpdateAverage(*state.Child(a), reach);
    return;
  }
  double total = std::accumulate(reach.begin(), reach.end(), 0.0);
  if (total == 0.0) return;
  Player cur = state.CurrentPlayer();
  std::string key = state.InformationStateString(cur);
  std::vector<Action> legal = state.LegalActions();
  auto ins = info_states_.insert({key, CFRInfoStateValues(legal, kInitialTableValues)});
  CFRInfoStateValues copy = ins.first->second;
  copy.ApplyRegretMatching();
  for (size_t a = 0; a < legal.size(); ++a) {
    std::vector<double> child_reach = reach;
    child_reach[cur] *= copy.current_policy[a];
    FullUpdateAverage(*state.Child(legal[a]), child_reach);
  }
  CFRInfoStateValues& row = info_states_[key];
  for (size_t a = 0; a < legal.size(); ++a)
    row.cumulative_policy[a] += reach[cur] * copy.current_policy[a];
}

// ============================================================================
// Outcome-sampling MCCFR (outcome_sampling_mccfr.cc)
// ============================================================================
OutcomeSamplingMCCFRSolver::OutcomeSamplingMCCFRSolver(const Game& game, double epsilon, int seed)
    : game_(game.shared_from_this()),
      epsilon_(epsilon),
      rng_(seed >= 0 ? seed : 0),
      dist_(0.0, 1.0),
      default_policy_(std::make_shared<UniformPolicy>()) {}

void OutcomeSamplingMCCFRSolver::RunIteration() {  // :67-74
  auto next_z = [this]() { return dist_(rng_); };
  for (Player p = 0; p < game_->NumPlayers(); ++p) {
    std::unique_ptr<State> state = game_->NewInitialState();
    SampleEpisodeWith(state.get(), p, next_z, 1.0, 1.0, 1.0);
  }
}

std::vector<double> OutcomeSamplingMCCFRSolver::SamplePolicy(const CFRInfoStateValues& info_state) const {
  std::vector<double> policy = info_state.current_policy;  // :111-118
  for (size_t i = 0; i < policy.size(); ++i)
    policy[i] = epsilon_ * 1.0 / policy.size() + (1 - epsilon_) * policy[i];
  return policy;
}

double OutcomeSamplingMCCFRSolver::SampleEpisodeWith(State* state, Player update_player,
                                                     const std::function<double()>& next_z, double my_reach,
                                                     double opp_reach, double sample_reach) {
  // outcome_sampling_mccfr.cc:141-241, Baseline() == 0 (vanilla outcome sampling, :120-124).
  if (state->IsTerminal()) return state->PlayerReturn(update_player);
  if (state->IsChanceNode()) {
    std::pair<Action, double> outcome = SampleAction(state->ChanceOutcomes(), next_z());
    state->ApplyAction(outcome.first);
    return SampleEpisodeWith(state, update_player, next_z, my_reach, outcome.second * opp_reach,
                             outcome.second * sample_reach);
  }
  const Player player = state->CurrentPlayer();
  const std::string key = state->InformationStateString(player);
  const std::vector<Action> legal = state->LegalActions();
  auto ins = info_states_.insert({key, CFRInfoStateValues(legal, kInitialTableValues)});
  CFRInfoStateValues copy = ins.first->second;
  copy.ApplyRegretMatching();
  const std::vector<double> sample_policy = player == update_player ? SamplePolicy(copy) : copy.current_policy;
  const double z = next_z();
  int sampled = static_cast<int>(legal.size()) - 1;
  double acc = 0.0;
  for (size_t a = 0; a < legal.size(); ++a) {
    if (z >= acc && z < acc + sample_policy[a]) { sampled = static_cast<int>(a); break; }
    acc += sample_policy[a];
  }
  state->ApplyAction(legal[sampled]);
  const double child_value = SampleEpisodeWith(
      state, update_player, next_z,
      player == update_player ? my_reach * copy.current_policy[sampled] : my_reach,
      player == update_player ? opp_reach : opp_reach * copy.current_policy[sampled],
      sample_reach * sample_policy[sampled]);
  std::vector<double> child_values(legal.size(), 0);
  for (size_t a = 0; a < legal.size(); ++a) {  // BaselineCorrectedChildValue with baseline 0 (:126-139)
    const double baseline = 0;
    child_values[a] = static_cast<int>(a) == sampled ? baseline + (child_value - baseline) / sample_policy[a] : baseline;
  }
  double value_estimate = 0;
  for (size_t a = 0; a < legal.size(); ++a) value_estimate += copy.current_policy[a] * child_values[a];
  if (player == update_player) {
    CFRInfoStateValues& row = info_states_[key];
    row.ApplyRegretMatching();
    const double cf_value = value_estimate * opp_reach / sample_reach;
    for (size_t a = 0; a < legal.size(); ++a) {
      const double cf_action_value = child_values[a] * opp_reach / sample_reach;
      row.cumulative_regrets[a] += (cf_action_value - cf_value);
    }
    for (size_t a = 0; a < legal.size(); ++a)
      row.cumulative_policy[a] += my_reach * row.current_policy[a] / sample_reach;
  }
  return value_estimate;
}

// ============================================================================
// The judge: expected returns, best response, NashConv, exploitability
// ============================================================================
static double ProbOf(const ActionsAndProbs& ap, Action a) {
  for (const auto& x : ap)
    if (x.first == a) return x.second;
  return -1.0;  // GetProb, policy.cc
}

std::vector<double> ExpectedReturns(const State& state, const Policy& policy) {
  // expected_returns.cc:34-130 (depth_limit=-1, prob_cut_threshold=0).
  if (state.IsTerminal()) return state.Returns();
  const int P = state.NumPlayers();
  std::vector<double> values(P, 0.0);
  if (state.IsChanceNode()) {
    for (const auto& ap : state.ChanceOutcomes()) {
      if (ap.second <= 0.0) continue;
      std::vector<double> cv = ExpectedReturns(*state.Child(ap.first), policy);
      for (int p = 0; p < P; ++p) values[p] += ap.second * cv[p];
    }
    return values;
  }
  ActionsAndProbs sp = policy.GetStatePolicy(state);  // use_state_get_policy
  if (sp.empty()) Fatal("ExpectedReturns: infostate not found.");
  for (Action a : state.LegalActions()) {
    double pr = ProbOf(sp, a);
    if (pr > 0.0) {
      std::vector<double> cv = ExpectedReturns(*state.Child(a), policy);
      for (int p = 0; p < P; ++p) values[p] += pr * cv[p];
    }
  }
  return values;
}

namespace {
// A flattened history tree for one responder (history_tree.cc +
// best_response.cc).  Node values are memoised by index; the responder's
// decision at an infostate maximises the counterfactual-reach-weighted sum
// over its member histories, ties -> lowest action (strict >).
struct BRNode {
  std::unique_ptr<State> state;
  int kind;  // 0 terminal, 1 chance, 2 decision
  std::string infostate;
  std::vector<Action> actions;
  std::vector<double> probs;  // chance / opponent-policy probs (1.0 for responder)
  std::vector<int> children;
  double cf_reach = 1.0;  // product of chance + opponents' probs on the path
  bool has_value = false;
  double value = 0;
};
struct BRTree {
  const Policy& policy;
  Player responder;
  std::vector<BRNode> nodes;
  std::unordered_map<std::string, std::vector<int>> infosets;
  std::unordered_map<std::string, Action> best_action;

  BRTree(const Game& game, Player r, const Policy& pol) : policy(pol), responder(r) {
    Build(game.NewInitialState(), 1.0);
  }
  int Build(std::unique_ptr<State> s, double cf) {
    int idx = static_cast<int>(nodes.size());
    nodes.emplace_back();
    nodes[idx].cf_reach = cf;
    std::vector<Action> acts;
    std::vector<double> probs;
    int kind;
    std::string key;
    if (s->IsTerminal()) {
      kind = 0;
    } else if (s->IsChanceNode()) {
      kind = 1;
      for (const auto& ap : s->ChanceOutcomes()) {
        acts.push_back(ap.first);
        probs.push_back(ap.second);
      }
    } else {
      kind = 2;
      key = s->InformationStateString();
      acts = s->LegalActions();
      if (s->CurrentPlayer() == responder) {
        probs.assign(acts.size(), 1.0);
        infosets[key].push_back(idx);
      } else {
        ActionsAndProbs sp = policy.GetStatePolicy(*s);
        if (sp.empty()) Fatal(key + " not found in policy.");
        for (Action a : acts) {
          double pr = ProbOf(sp, a);
          probs.push_back(pr < 0 ? 0.0 : pr);
        }
      }
    }
    std::vector<int> kids;
    for (size_t i = 0; i < acts.size(); ++i)
      kids.push_back(Build(s->Child(acts[i]), cf * probs[i]));
    BRNode& n = nodes[idx];
    n.kind = kind;
    n.infostate = key;
    n.actions = acts;
    n.probs = probs;
    n.children = kids;
    n.state = std::move(s);
    return idx;
  }
  Action BestAction(const std::string& key) {  // best_response.cc:194-227
    auto it = best_action.find(key);
    if (it != best_action.end()) return it->second;
    const std::vector<int>& members = infosets.at(key);
    Action best = -1;
    double best_v = std::numeric_limits<double>::lowest();
    const BRNode& first = nodes[members[0]];
    for (size_t ai = 0; ai < first.actions.size(); ++ai) {
      double v = 0;
      for (int m : members) {
        if (nodes[m].cf_reach <= -1.0) continue;  // prob_cut_threshold = -1
        v += nodes[m].cf_reach * Value(nodes[m].children[ai]);
      }
      if (v > best_v) {
        best_v = v;
        best = first.actions[ai];
      }
    }
    if (best == -1) Fatal("No action was chosen.");
    best_action[key] = best;
    return best;
  }
  double Value(int idx) {
    if (nodes[idx].has_value) return nodes[idx].value;
    double v = 0;
    BRNode& n = nodes[idx];
    if (n.kind == 0) {
      v = n.state->Returns()[responder];
    } else if (n.kind == 1) {
      for (size_t i = 0; i < n.actions.size(); ++i) v += n.probs[i] * Value(n.children[i]);
    } else if (n.state->CurrentPlayer() == responder) {
      Action a = BestAction(n.infostate);
      for (size_t i = 0; i < n.actions.size(); ++i)
        if (nodes[idx].actions[i] == a) v += 1.0 * Value(nodes[idx].children[i]);
    } else {
      for (size_t i = 0; i < n.actions.size(); ++i) {
        if (nodes[idx].probs[i] <= -1.0) continue;
        v += nodes[idx].probs[i] * Value(nodes[idx].children[i]);
      }
    }
    nodes[idx].has_value = true;
    nodes[idx].value = v;
    return v;
  }
};
}  // namespace

double BestResponseValue(const Game& game, Player responder, const Policy& policy) {
  BRTree tree(game, responder, policy);
  return tree.Value(0);
}
std::unordered_map<std::string, Action> BestResponseActions(const Game& game, Player responder, const Policy& policy) {
  BRTree tree(game, responder, policy);
  tree.Value(0);  // fills the cache "starting at the root" (best_response.h:111-114)
  for (const auto& kv : tree.infosets) tree.BestAction(kv.first);
  return tree.best_action;
}

void CFRBRSolver::EvaluateAndUpdatePolicy() {  // cfr_br.cc:48-83
  ++iteration_;
  const int P = game_->NumPlayers();
  // iteration 1 responds to the uniform policy (cfr_br.cc:59-61) — which is what the current policy of a fresh
  // table is, row for row
  std::shared_ptr<Policy> current = CurrentPolicy();
  std::vector<std::unordered_map<std::string, Action>> br(P);
  for (int p = 0; p < P; ++p) br[p] = BestResponseActions(*game_, p, *current);
  std::vector<const std::unordered_map<std::string, Action>*> overrides(P, nullptr);
  overrides_ = &overrides;
  for (int p = 0; p < P; ++p) {
    for (int opp = 0; opp < P; ++opp) overrides[opp] = opp == p ? nullptr : &br[opp];
    ComputeCounterFactualRegret(*root_state_, p, root_reach_probs_);
  }
  overrides_ = nullptr;
  ApplyRegretMatching();
}
double NashConv(const Game& game, const Policy& policy) {
  // tabular_exploitability.cc:60-89
  std::unique_ptr<State> root = game.NewInitialState();
  std::vector<double> on_policy = ExpectedReturns(*root, policy);
  double total = 0;
  for (Player p = 0; p < game.NumPlayers(); ++p) {
    double incentive = BestResponseValue(game, p, policy) - on_policy[p];
    if (incentive < -1e-9) Fatal("Negative Nash deviation incentive");
    total += incentive;
  }
  return total;
}
double Exploitability(const Game& game, const Policy& policy) {
  // tabular_exploitability.cc:30-47
  double total = 0;
  for (Player p = 0; p < game.NumPlayers(); ++p) total += BestResponseValue(game, p, policy);
  return (total - game.UtilitySum()) / game.NumPlayers();
}

}  // namespace osg_oracle
