// =============================================================================
// CPU ORACLE — TEST INFRASTRUCTURE ONLY.
//
// A dependency-free C++17 restatement of the OpenSpiel hot path (State/Game
// semantics for tic_tac_toe, connect_four, hex, kuhn_poker, leduc_poker; the
// random-rollout evaluator + MCTS; tabular CFR / CFR+ / external-sampling
// MCCFR; exploitability as the CFR judge).  It exists to CHECK the HIP path and
// to provide the `cpu_baseline` leg of bench.py.  Nothing in the product
// (open_spiel_amd/) may include, link or call it: only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() do.
//
// "Reference-shaped" on purpose: virtual State, std::vector return values,
// Clone() per rollout / tree edge, string-keyed unordered_map CFR tables — the
// same data-structure shapes as the reference, so timing it is an honest CPU
// baseline ("kind": "port"; bench.py prefers the genuine build, "kind": "reference").
//
// Parity pin: this restatement is compared CALL FOR CALL with the genuine reference
// implementation — the reference's own .cc files compiled unmodified from
// /root/reference into oracle/_ref/libspiel_ref.so (oracle/Makefile.ref, with the
// private abseil / nlohmann stand-ins of oracle/ref_shim) behind the same extern "C"
// driver (spiel_oracle_capi.cpp -DOSGO_GENUINE_REFERENCE): seeded playouts, strings,
// CFR / CFR+ tables, ES-MCCFR / OS-MCCFR tables, MCTSBot search trees and the
// exploitability judge are bit-identical (tests/test_oracle_vs_reference.py), and the
// outputs of that reference build are committed as golden vectors
// (tests/golden/reference_vectors.npz).  It is also pinned by the reference's playthrough
// goldens and known-answer tests (tests/golden/playthroughs.json, tests/test_oracle_*.py).
// What stays UNPINNED is abseil's random streams (absl::Uniform in mcts.cc:54 /
// spiel.cc:368-371, absl::discrete_distribution in outcome_sampling_mccfr.cc:178):
// abseil is not vendored in /root/reference and its distributions are unspecified; the
// stand-in draws from libstdc++'s engines with the conventions used here.  ES-MCCFR uses
// only std::mt19937 + std::uniform_real_distribution and IS stream-identical.
//
// Each function cites the reference file:line it restates (paths relative to
// /root/reference/open_spiel/).
// =============================================================================
#ifndef OSG_ORACLE_SPIEL_ORACLE_H_
#define OSG_ORACLE_SPIEL_ORACLE_H_

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace osg_oracle {

// spiel_utils.h:134-135
using Action = int64_t;
using Player = int;
// spiel_globals.h:26-56,82
constexpr Player kChancePlayerId = -1;
constexpr Player kSimultaneousPlayerId = -2;
constexpr Player kInvalidPlayer = -3;
constexpr Player kTerminalPlayerId = -4;
constexpr Action kInvalidAction = -1;

using ActionsAndProbs = std::vector<std::pair<Action, double>>;  // spiel.h:224

// Fatal errors become exceptions (the reference's pybind handler does the same,
// python/pybind11/pyspiel.cc:831-837).
struct SpielError : public std::runtime_error {
  using std::runtime_error::runtime_error;
};
[[noreturn]] void Fatal(const std::string& msg);
#define ORACLE_CHECK(cond)                                                   \
  do {                                                                       \
    if (!(cond))                                                             \
      ::osg_oracle::Fatal(std::string(__FILE__) + ":" +                      \
                          std::to_string(__LINE__) + " CHECK failed: " #cond); \
  } while (0)

// game_parameters.h:40 / game_parameters.cc:172-227 — "name(k=v,k=v)".
struct ParamValue {
  enum Kind { kBool, kInt, kDouble, kString } kind = kString;
  bool b = false;
  int i = 0;
  double d = 0;
  std::string s;
  std::string ToString() const;
};
using GameParams = std::map<std::string, ParamValue>;
GameParams ParseGameString(const std::string& game_string);

class Game;

struct PlayerAction {  // spiel.h State::PlayerAction
  Player player;
  Action action;
};

// spiel.h:301-916 (the subset on the hot path).
class State {
 public:
  explicit State(std::shared_ptr<const Game> game);
  virtual ~State() = default;

  virtual Player CurrentPlayer() const = 0;
  virtual std::vector<Action> LegalActions() const = 0;
  virtual std::string ActionToString(Player player, Action a) const = 0;
  virtual std::string ToString() const = 0;
  virtual bool IsTerminal() const = 0;
  virtual std::vector<double> Returns() const = 0;
  virtual std::unique_ptr<State> Clone() const = 0;
  virtual ActionsAndProbs ChanceOutcomes() const;
  virtual std::string InformationStateString(Player player) const;
  virtual std::string ObservationString(Player player) const;
  virtual void InformationStateTensor(Player player, float* out, int n) const;
  virtual void ObservationTensor(Player player, float* out, int n) const;

  // spiel.cc:441-451
  void ApplyAction(Action a);
  // spiel.h:366-372
  std::vector<Action> LegalActions(Player player) const;
  // spiel.cc:518-524
  std::vector<int> LegalActionsMask(Player player) const;
  std::vector<int> LegalActionsMask() const {
    return LegalActionsMask(CurrentPlayer());
  }
  bool IsChanceNode() const { return CurrentPlayer() == kChancePlayerId; }
  // Terminal-reward games: Rewards() == Returns() at terminal, zeros before
  // (spiel.h:421-437; every game here has RewardModel::kTerminal).
  std::vector<double> Rewards() const { return Returns(); }
  double PlayerReturn(Player p) const { return Returns()[p]; }
  std::unique_ptr<State> Child(Action a) const {  // spiel.h:737-744
    auto c = Clone();
    c->ApplyAction(a);
    return c;
  }
  std::string InformationStateString() const {
    return InformationStateString(CurrentPlayer());
  }
  // spiel.cc:908-945 (bounds-checked wrappers returning by value).
  std::vector<float> ObservationTensor(Player player) const;
  std::vector<float> InformationStateTensor(Player player) const;

  std::vector<Action> History() const;
  const std::vector<PlayerAction>& FullHistory() const { return history_; }
  std::string HistoryString() const;  // "a, b, c"
  int MoveNumber() const { return move_number_; }
  int NumPlayers() const { return num_players_; }
  int NumDistinctActions() const { return num_distinct_actions_; }
  std::shared_ptr<const Game> GetGame() const { return game_; }

 protected:
  virtual void DoApplyAction(Action a) = 0;
  std::shared_ptr<const Game> game_;
  int num_distinct_actions_;
  int num_players_;
  std::vector<PlayerAction> history_;
  int move_number_ = 0;
};

// spiel.h:927-1255 (subset).
class Game : public std::enable_shared_from_this<Game> {
 public:
  Game(std::string short_name, GameParams params)
      : short_name_(std::move(short_name)), given_(params), params_(std::move(params)) {}
  virtual ~Game() = default;
  virtual int NumDistinctActions() const = 0;
  virtual std::unique_ptr<State> NewInitialState() const = 0;
  virtual int MaxChanceOutcomes() const { return 0; }
  virtual int NumPlayers() const = 0;
  virtual double MinUtility() const = 0;
  virtual double MaxUtility() const = 0;
  virtual double UtilitySum() const { return 0; }
  virtual std::vector<int> ObservationTensorShape() const { return {}; }
  virtual std::vector<int> InformationStateTensorShape() const { return {}; }
  virtual int MaxGameLength() const = 0;
  virtual int MaxChanceNodesInHistory() const { return 0; }
  virtual bool HasChance() const { return false; }
  int ObservationTensorSize() const { return Product(ObservationTensorShape()); }
  int InformationStateTensorSize() const {
    return Product(InformationStateTensorShape());
  }
  const std::string& ShortName() const { return short_name_; }
  const GameParams& Params() const { return params_; }
  std::string ToString() const;           // "name(k=v,...)": parameters as given
  std::string ParametersString() const;   // "{k=v,...}" incl. defaults

 protected:
  static int Product(const std::vector<int>& s) {
    if (s.empty()) return 0;
    int p = 1;
    for (int v : s) p *= v;
    return p;
  }
  bool BoolParam(const std::string& k, bool def);
  int IntParam(const std::string& k, int def);
  std::string StrParam(const std::string& k, const std::string& def);
  std::string short_name_;
  GameParams given_;   // exactly what the game string carried
  GameParams params_;  // after the ctor: every parameter incl. defaults
};

// spiel.cc:255 LoadGame — only the five hot-path games are registered.
std::shared_ptr<const Game> LoadGame(const std::string& game_string);

// spiel.cc:372-409
std::pair<Action, double> SampleAction(const ActionsAndProbs& outcomes,
                                       double z);

// ---------------------------------------------------------------------------
// Counter-based RNG shared (bit for bit) with the HIP kernels, so that device
// rollouts / searches / MCCFR trajectories can be REPLAYED here.  Not part of
// the reference (which uses mt19937 + absl distributions; stream unpinned).
// ---------------------------------------------------------------------------
struct CounterRng {
  uint64_t s;
  explicit CounterRng(uint64_t seed, uint64_t stream = 0, uint64_t sub = 0);
  uint64_t Next();                 // splitmix64 step
  uint32_t Below(uint32_t n);      // floor((hi32 * n) / 2^32)
  double Unit();                   // (top 53 bits) * 2^-53, in [0,1)
};

// Keyed orderings of the device's wave-per-root search (open_spiel_amd/csrc/
// osg_common.h: order_key / fill_key), restated bit for bit.  Not in the reference.
uint64_t Mix64(uint64_t z);
uint64_t PathHashRoot();
uint64_t PathHashChild(uint64_t parent, int action);
uint64_t OrderBase(uint64_t seed, uint64_t root);
uint32_t OrderKey(uint64_t base, uint64_t parent_path_hash, int action);
uint64_t FillBase(uint64_t seed, uint64_t root, uint64_t sub);
uint64_t FillKey(uint64_t base, int cell);

// ---------------------------------------------------------------------------
// Policies (policy.h:69,158,318 subset)
// ---------------------------------------------------------------------------
class Policy {
 public:
  virtual ~Policy() = default;
  virtual ActionsAndProbs GetStatePolicy(const State& state, Player p) const {
    return GetStatePolicy(state.InformationStateString(p));
  }
  ActionsAndProbs GetStatePolicy(const State& state) const {
    return GetStatePolicy(state, state.CurrentPlayer());
  }
  virtual ActionsAndProbs GetStatePolicy(const std::string& info_state) const {
    (void)info_state;
    Fatal("GetStatePolicy(string) unimplemented");
  }
};
class TabularPolicy : public Policy {
 public:
  TabularPolicy() = default;
  explicit TabularPolicy(std::unordered_map<std::string, ActionsAndProbs> t)
      : table_(std::move(t)) {}
  using Policy::GetStatePolicy;
  ActionsAndProbs GetStatePolicy(const std::string& info_state) const override {
    auto it = table_.find(info_state);
    return it == table_.end() ? ActionsAndProbs{} : it->second;
  }
  std::unordered_map<std::string, ActionsAndProbs>& Table() { return table_; }
  const std::unordered_map<std::string, ActionsAndProbs>& Table() const {
    return table_;
  }

 private:
  std::unordered_map<std::string, ActionsAndProbs> table_;
};
class UniformPolicy : public Policy {
 public:
  using Policy::GetStatePolicy;
  ActionsAndProbs GetStatePolicy(const State& state, Player p) const override;
};
TabularPolicy GetUniformPolicy(const Game& game);      // policy.h:390
TabularPolicy GetFirstActionPolicy(const Game& game);  // policy.cc
TabularPolicy KuhnOptimalPolicy(double alpha);         // kuhn_poker.cc:451-474

// ---------------------------------------------------------------------------
// MCTS (algorithms/mcts.{h,cc})
// ---------------------------------------------------------------------------
class Evaluator {  // mcts.h:83-92
 public:
  virtual ~Evaluator() = default;
  virtual std::vector<double> Evaluate(const State& state) = 0;
  virtual ActionsAndProbs Prior(const State& state) = 0;
};
class RandomRolloutEvaluator : public Evaluator {  // mcts.h:97-111
 public:
  RandomRolloutEvaluator(int n_rollouts, int seed)
      : n_rollouts_(n_rollouts), rng_(seed) {}
  std::vector<double> Evaluate(const State& state) override;  // mcts.cc:43-72
  ActionsAndProbs Prior(const State& state) override;         // mcts.cc:74-87

 private:
  int n_rollouts_;
  std::mt19937 rng_;
};
struct SearchNode {  // mcts.h:114-146
  Action action = 0;
  double prior = 0;
  Player player = 0;
  int explore_count = 0;
  double total_reward = 0;
  std::vector<double> outcome;
  std::vector<SearchNode> children;
  SearchNode() = default;
  SearchNode(Action a, Player p, double pr) : action(a), prior(pr), player(p) {}
  double UCTValue(int parent_explore_count, double uct_c) const;
  double PUCTValue(int parent_explore_count, double uct_c) const;
  bool CompareFinal(const SearchNode& b) const;
  const SearchNode& BestChild() const;
};
enum class ChildSelectionPolicy { UCT, PUCT };
class MCTSBot {  // mcts.h:149-220
 public:
  MCTSBot(const Game& game, std::shared_ptr<Evaluator> evaluator, double uct_c,
          int max_simulations, int64_t max_memory_mb, bool solve, int seed,
          bool verbose,
          ChildSelectionPolicy policy = ChildSelectionPolicy::UCT,
          bool dont_return_chance_node = false);
  Action Step(const State& state);
  std::unique_ptr<SearchNode> MCTSearch(const State& state);
  int LastNodeCount() const { return nodes_; }
  // Replay mode (NOT in the reference): every random draw comes from the counter
  // streams the HIP search uses for root `root_index`, so a device search can be
  // reproduced node for node.  Simulation s shuffles / samples chance with
  // CounterRng(seed ^ kTreeSalt, root_index, s); rollout r of simulation s plays
  // with CounterRng(seed, root_index, s * n_rollouts + r).
  static constexpr uint64_t kTreeSalt = 0x7265655F73616C74ULL;
  //
  // layout 1 = the device's lane-per-root kernel (above).  layout 2 = its
  // wave-per-root kernel: a new node's children are ordered by OrderKey (instead of
  // shuffled) and, for hex without the swap rule, a rollout plays the empty cells in
  // the interleaved FillKey order (mover: the ceil(m/2) smallest keys ascending,
  // opponent: the others ascending) — a uniformly random move sequence.
  void UseCounterStreams(uint64_t seed, uint64_t root_index, int n_rollouts, int layout = 1) {
    counter_ = true;
    c_seed_ = seed;
    c_root_ = root_index;
    c_rollouts_ = n_rollouts;
    c_layout_ = layout;
  }
  // Replay mode with a real evaluator: leaves go to evaluator_->Evaluate instead of the counter-stream rollouts
  // (the device's evaluator-outside-the-kernel search, osg_mcts_tree_*).
  void UseEvaluatorInReplay() { c_use_evaluator_ = true; }

 private:
  std::unique_ptr<State> ApplyTreePolicy(SearchNode* root, const State& state,
                                         std::vector<SearchNode*>* visit_path);
  void GarbageCollect(SearchNode* node);
  bool c_use_evaluator_ = false;
  double uct_c_;
  int max_simulations_;
  int max_nodes_;
  int nodes_ = 0;
  int gc_limit_;
  bool solve_;
  double max_utility_;
  bool dont_return_chance_node_;
  std::mt19937 rng_;
  ChildSelectionPolicy child_selection_policy_;
  std::shared_ptr<Evaluator> evaluator_;
  std::vector<double> CounterEvaluate(const State& state, int sim) const;
  bool counter_ = false;
  uint64_t c_seed_ = 0, c_root_ = 0;
  int c_rollouts_ = 1;
  int c_layout_ = 1;
  CounterRng trng_{0};
};

// ---------------------------------------------------------------------------
// CFR family (algorithms/cfr.{h,cc}, external_sampling_mccfr.{h,cc})
// ---------------------------------------------------------------------------
struct CFRInfoStateValues {  // cfr.h:42-98
  CFRInfoStateValues() = default;
  CFRInfoStateValues(std::vector<Action> la, double init_value)
      : legal_actions(std::move(la)),
        cumulative_regrets(legal_actions.size(), init_value),
        cumulative_policy(legal_actions.size(), init_value),
        current_policy(legal_actions.size(), 1.0 / legal_actions.size()) {}
  explicit CFRInfoStateValues(std::vector<Action> la)
      : CFRInfoStateValues(std::move(la), 0) {}
  void ApplyRegretMatching();                       // cfr.cc:596-615
  int SampleActionIndex(double epsilon, double z);  // cfr.cc:617-628
  int num_actions() const { return static_cast<int>(legal_actions.size()); }
  bool empty() const { return legal_actions.empty(); }
  std::vector<Action> legal_actions;
  std::vector<double> cumulative_regrets;
  std::vector<double> cumulative_policy;
  std::vector<double> current_policy;
};
using CFRInfoStateValuesTable =
    std::unordered_map<std::string, CFRInfoStateValues>;  // cfr.h:103-104

class CFRAveragePolicy : public Policy {  // cfr.h:122-147, cfr.cc:68-135
 public:
  CFRAveragePolicy(const CFRInfoStateValuesTable& t,
                   std::shared_ptr<Policy> default_policy)
      : info_states_(t), default_policy_(std::move(default_policy)) {}
  using Policy::GetStatePolicy;
  ActionsAndProbs GetStatePolicy(const State& state, Player p) const override;
  ActionsAndProbs GetStatePolicy(const std::string& info_state) const override;
  TabularPolicy AsTabular() const;
  static ActionsAndProbs FromValues(const CFRInfoStateValues& v);

 private:
  const CFRInfoStateValuesTable& info_states_;
  std::shared_ptr<Policy> default_policy_;
};
class CFRCurrentPolicy : public Policy {  // cfr.h:150-172
 public:
  explicit CFRCurrentPolicy(const CFRInfoStateValuesTable& t)
      : info_states_(t) {}
  using Policy::GetStatePolicy;
  ActionsAndProbs GetStatePolicy(const std::string& info_state) const override;

 private:
  const CFRInfoStateValuesTable& info_states_;
};

class CFRSolverBase {  // cfr.h:188-304
 public:
  CFRSolverBase(const Game& game, bool alternating_updates,
                bool linear_averaging, bool regret_matching_plus);
  virtual ~CFRSolverBase() = default;
  virtual void EvaluateAndUpdatePolicy();  // cfr.cc:263-282
  std::shared_ptr<Policy> AveragePolicy() const {
    return std::make_shared<CFRAveragePolicy>(info_states_, nullptr);
  }
  std::shared_ptr<Policy> CurrentPolicy() const {
    return std::make_shared<CFRCurrentPolicy>(info_states_);
  }
  CFRInfoStateValuesTable& InfoStateValuesTable() { return info_states_; }
  int Iteration() const { return iteration_; }

 protected:
  std::vector<double> ComputeCounterFactualRegret(
      const State& state, int alternating_player /* -1 = all */,
      const std::vector<double>& reach_probabilities);
  std::vector<double> ForActionProbs(const State& state, int alternating_player,
                                     const std::vector<double>& reach,
                                     int current_player,
                                     const std::vector<double>& probs,
                                     const std::vector<Action>& actions,
                                     std::vector<double>* child_values_out);
  void InitializeInfostateNodes(const State& state);
  void ApplyRegretMatching();
  void ApplyRegretMatchingPlusReset();
  std::shared_ptr<const Game> game_;
  int iteration_ = 0;
  CFRInfoStateValuesTable info_states_;
  std::unique_ptr<State> root_state_;
  std::vector<double> root_reach_probs_;
  bool regret_matching_plus_, alternating_updates_, linear_averaging_;
  int chance_player_;
  // policy_overrides (cfr.cc:331-372): per player, the deterministic best-response action of every one of its
  // infostates, or null = use the current policy.  Set by CFRBRSolver only.
  const std::vector<const std::unordered_map<std::string, Action>*>* overrides_ = nullptr;
};
// cfr_br.h:34-56, cfr_br.cc:23-83: every player minimises regret against the others' best responses to
// the current policy.
class CFRBRSolver : public CFRSolverBase {
 public:
  explicit CFRBRSolver(const Game& game) : CFRSolverBase(game, false, false, false) {}
  void EvaluateAndUpdatePolicy();
};
// TabularBestResponse::GetBestResponseActions (best_response.h:106-115) of `responder` against `policy`.
std::unordered_map<std::string, Action> BestResponseActions(const Game& game, Player responder, const Policy& policy);
class CFRSolver : public CFRSolverBase {  // cfr.h:310-330
 public:
  explicit CFRSolver(const Game& game)
      : CFRSolverBase(game, true, false, false) {}
};
class CFRPlusSolver : public CFRSolverBase {  // cfr.h:341-357
 public:
  explicit CFRPlusSolver(const Game& game)
      : CFRSolverBase(game, true, true, true) {}
};

enum class AverageType { kSimple, kFull };
class ExternalSamplingMCCFRSolver {  // external_sampling_mccfr.h:57-113
 public:
  static constexpr double kInitialTableValues = 0.000001;
  ExternalSamplingMCCFRSolver(const Game& game, int seed = 0,
                              AverageType avg = AverageType::kSimple);
  void RunIteration();
  // One traverser pass driven by an explicit uniform source (used to replay a
  // device trajectory: "same table + same z-sequence => same deltas").
  // on_branch (the device's stream rule, csrc/osg_cfr.hip): called as (level, b1, b2) before the traversal enters child
  // b1 of the FIRST node at which `player` acts on the trajectory (level 1) and child b2 of the SECOND such node on the
  // path inside child b1 (level 2) — the caller switches its draw source to that subtree's own stream; null: one stream
  // in visiting order, as in the reference.
  double UpdateRegretsWith(const State& state, Player player, const std::function<double()>& next_z,
                           const std::function<void(int, int, int)>* on_branch = nullptr, int depth = 0, int b1 = 0);
  CFRInfoStateValuesTable& InfoStateValuesTable() { return info_states_; }
  std::shared_ptr<Policy> AveragePolicy() const {
    return std::make_shared<CFRAveragePolicy>(info_states_, default_policy_);
  }
  // the second half of a kFull RunIteration (external_sampling_mccfr.cc:76-79) on its own: replay hook
  void FullUpdateAverageFromRoot() {
    FullUpdateAverage(*game_->NewInitialState(), std::vector<double>(game_->NumPlayers(), 1.0));
  }

 private:
  void FullUpdateAverage(const State& state, const std::vector<double>& reach);
  std::shared_ptr<const Game> game_;
  std::mt19937 rng_;
  AverageType avg_type_;
  CFRInfoStateValuesTable info_states_;
  std::uniform_real_distribution<double> dist_;
  std::shared_ptr<Policy> default_policy_;
};

class OutcomeSamplingMCCFRSolver {  // outcome_sampling_mccfr.h:40-107
 public:
  static constexpr double kInitialTableValues = 0.000001;
  static constexpr double kDefaultEpsilon = 0.6;
  OutcomeSamplingMCCFRSolver(const Game& game, double epsilon = kDefaultEpsilon, int seed = -1);
  void RunIteration();  // one SampleEpisode per player (outcome_sampling_mccfr.cc:67-74)
  // One episode driven by an explicit uniform source (replay of a device trajectory).  The
  // reference draws the action from absl::discrete_distribution (stream unpinned); here it is
  // the first index whose cumulative sample probability exceeds z.
  double SampleEpisodeWith(State* state, Player update_player, const std::function<double()>& next_z,
                           double my_reach, double opp_reach, double sample_reach);
  CFRInfoStateValuesTable& InfoStateValuesTable() { return info_states_; }
  std::shared_ptr<Policy> AveragePolicy() const {
    return std::make_shared<CFRAveragePolicy>(info_states_, default_policy_);
  }

 private:
  std::vector<double> SamplePolicy(const CFRInfoStateValues& info_state) const;  // :111-118
  std::shared_ptr<const Game> game_;
  double epsilon_;
  CFRInfoStateValuesTable info_states_;
  std::mt19937 rng_;
  std::uniform_real_distribution<double> dist_;
  std::shared_ptr<Policy> default_policy_;
};

// ---------------------------------------------------------------------------
// The CFR judge (algorithms/expected_returns.cc, best_response.cc,
// tabular_exploitability.cc)
// ---------------------------------------------------------------------------
std::vector<double> ExpectedReturns(const State& state, const Policy& policy);
double BestResponseValue(const Game& game, Player responder,
                         const Policy& policy);
double NashConv(const Game& game, const Policy& policy);
double Exploitability(const Game& game, const Policy& policy);

}  // namespace osg_oracle

#endif  // OSG_ORACLE_SPIEL_ORACLE_H_
