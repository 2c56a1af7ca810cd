// Source-level drop-in check: ONE user program, written against the reference's C++ API
// in the style of open_spiel/examples/{example,mcts_example,cfr_example}.cc, compiled twice FROM THE SAME TEXT — no
// #ifdef, no namespace switch; only the include path and the library differ:
//   * -I /root/reference (+ the abseil stand-ins):  the GENUINE reference headers, linked with oracle/_ref/libspiel_ref.so
//   * -I <repo>/include:                            the MI355X drop-in headers (include/open_spiel/** -> the host
//                                                   mirror over the C-ABI), linked with open_spiel_amd/libosg_hip.so
// Both print the same transcript; tests/test_dropin.py builds both (CPU), runs the reference build and
// keeps its transcript, and tests/test_z2_gpu_dropin.py runs the mirror build on the GPU and compares.
// Every line is deterministic: fixed move choices, the full-tree CFR family, MCTS-Solver proofs.
#include <cstdio>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "open_spiel/algorithms/best_response.h"
#include "open_spiel/algorithms/cfr.h"
#include "open_spiel/algorithms/cfr_br.h"
#include "open_spiel/algorithms/expected_returns.h"
#include "open_spiel/algorithms/external_sampling_mccfr.h"
#include "open_spiel/algorithms/mcts.h"
#include "open_spiel/policy.h"
#include "open_spiel/spiel_bots.h"
#include "open_spiel/algorithms/tabular_exploitability.h"
#include "open_spiel/spiel.h"
namespace spiel = open_spiel;
namespace algos = open_spiel::algorithms;

using spiel::Action;

static void PrintVector(const char* label, const std::vector<double>& v) {
  std::printf("%s", label);
  for (double x : v) std::printf(" %.12g", x);
  std::printf("\n");
}

// A fixed playout: the mover takes legal[(7 * ply + 3) % n], chance takes outcome (5 * ply + 1) % n.
static void Playthrough(const std::string& game_string) {
  std::shared_ptr<const spiel::Game> game = spiel::LoadGame(game_string);
  std::printf("game %s actions=%d players=%d obs=%d maxlen=%d utility=[%g,%g]\n", game->ToString().c_str(),
              game->NumDistinctActions(), game->NumPlayers(), game->ObservationTensorSize(), game->MaxGameLength(),
              game->MinUtility(), game->MaxUtility());
  std::unique_ptr<spiel::State> state = game->NewInitialState();
  int ply = 0;
  while (!state->IsTerminal()) {
    Action action;
    if (state->IsChanceNode()) {
      spiel::ActionsAndProbs outcomes = state->ChanceOutcomes();
      std::printf("  ply %d chance:", ply);
      for (const auto& ap : outcomes) std::printf(" %d@%.6f", static_cast<int>(ap.first), ap.second);
      action = outcomes[(5 * ply + 1) % outcomes.size()].first;
    } else {
      std::vector<Action> legal = state->LegalActions();
      std::printf("  ply %d player %d legal:", ply, state->CurrentPlayer());
      for (Action a : legal) std::printf(" %d", static_cast<int>(a));
      action = legal[(7 * ply + 3) % legal.size()];
      for (int p = 0; p < game->NumPlayers(); ++p) {
        std::vector<float> obs = state->ObservationTensor(p);
        double sum = 0, weighted = 0;
        for (size_t i = 0; i < obs.size(); ++i) { sum += obs[i]; weighted += obs[i] * static_cast<double>(i % 97); }
        std::printf(" | obs%d %g/%g", p, sum, weighted);
      }
      std::printf(" | %s", state->ActionToString(state->CurrentPlayer(), action).c_str());
    }
    std::printf(" -> %d\n", static_cast<int>(action));
    std::unique_ptr<spiel::State> child = state->Child(action);   // Child() and ApplyAction() agree
    state->ApplyAction(action);
    if (child->HistoryString() != state->HistoryString()) std::printf("  CHILD MISMATCH\n");
    ++ply;
  }
  std::printf("  history %s\n", state->HistoryString().c_str());
  std::printf("  final\n%s\n", state->ToString().c_str());
  PrintVector("  returns", state->Returns());
  std::unique_ptr<spiel::State> copy = state->Clone();
  std::printf("  clone terminal=%d player=%d\n", copy->IsTerminal() ? 1 : 0, copy->CurrentPlayer());
}

static void InformationStates() {
  for (const char* name : {"kuhn_poker", "leduc_poker"}) {
    std::shared_ptr<const spiel::Game> game = spiel::LoadGame(name);
    std::unique_ptr<spiel::State> state = game->NewInitialState();
    for (Action a : std::vector<Action>{1, 2, 1, 1}) {
      if (state->IsTerminal()) break;
      bool ok = false;
      for (Action l : state->LegalActions()) ok |= l == a;
      state->ApplyAction(ok ? a : state->LegalActions()[0]);
    }
    for (int p = 0; p < game->NumPlayers(); ++p)
      std::printf("%s p%d info '%s' obs '%s'\n", name, p, state->InformationStateString(p).c_str(),
                  state->ObservationString(p).c_str());
  }
}

// mcts_test.cc:126-155: MCTS-Solver proofs on tic_tac_toe (the answers do not depend on the random streams).
static void SolverAnswers() {
  std::shared_ptr<const spiel::Game> game = spiel::LoadGame("tic_tac_toe");
  struct Case { std::vector<Action> moves; const char* name; };
  const std::vector<Case> cases = {{{4, 0, 8}, "x(1,1) o(0,0) x(2,2): draw"}, {{4, 0, 8, 1, 2}, "... o(0,1) x(0,2): o has lost"},
                                   {{1, 8}, "x(0,1) o(2,2): x wins"}};
  for (const Case& c : cases) {
    std::unique_ptr<spiel::State> state = game->NewInitialState();
    for (Action a : c.moves) state->ApplyAction(a);
    auto evaluator = std::make_shared<algos::RandomRolloutEvaluator>(/*n_rollouts=*/20, /*seed=*/42);
    algos::MCTSBot bot(*game, evaluator, /*uct_c=*/2.0, /*max_simulations=*/10000, /*max_memory_mb=*/10,
                       /*solve=*/true, /*seed=*/42, /*verbose=*/false);
    std::unique_ptr<algos::SearchNode> root = bot.MCTSearch(*state);
    const algos::SearchNode& best = root->BestChild();
    std::printf("solver '%s': proven=%d", c.name, root->outcome.empty() ? 0 : 1);
    if (!root->outcome.empty()) std::printf(" value %g", root->outcome[state->CurrentPlayer()]);
    if (!best.outcome.empty()) std::printf(" best-child value %g", best.outcome[state->CurrentPlayer()]);
    std::printf("\n");
  }
}

// cfr_example.cc:41-45 / cfr_test.cc:36-62.
static void Cfr() {
  std::shared_ptr<const spiel::Game> game = spiel::LoadGame("kuhn_poker");
  algos::CFRSolver solver(*game);
  for (int i = 1; i <= 100; ++i) {
    solver.EvaluateAndUpdatePolicy();
    if (i == 1 || i == 10 || i == 100)
      std::printf("kuhn CFR iter %d exploitability %.10f nash_conv %.10f\n", i,
                  algos::Exploitability(*game, solver.TabularAveragePolicy()),
                  algos::NashConv(*game, solver.TabularAveragePolicy()));
  }
  algos::CFRPlusSolver plus(*game);
  for (int i = 0; i < 50; ++i) plus.EvaluateAndUpdatePolicy();
  std::printf("kuhn CFR+ iter 50 exploitability %.10f\n", algos::Exploitability(*game, plus.TabularAveragePolicy()));
  std::shared_ptr<const spiel::Game> leduc = spiel::LoadGame("leduc_poker");
  algos::CFRSolver big(*leduc);
  for (int i = 0; i < 3; ++i) big.EvaluateAndUpdatePolicy();
  std::printf("leduc CFR iter 3 nash_conv %.10f infostates %d\n", algos::NashConv(*leduc, big.TabularAveragePolicy()),
              static_cast<int>(big.InfoStateValuesTable().size()));
}

// Round 2: the policy hierarchy, the best response, CFR-BR, external-sampling MCCFR driven by the caller's
// std::mt19937 (both average types), and MCTSBot through the Bot interface.
static void PoliciesAndSiblingSolvers() {
  std::shared_ptr<const spiel::Game> game = spiel::LoadGame("kuhn_poker");
  spiel::UniformPolicy uniform;
  spiel::TabularPolicy first_action = spiel::GetFirstActionPolicy(*game);
  spiel::TabularPolicy uniform_table = spiel::GetUniformPolicy(*game);
  std::printf("kuhn policies: uniform nash_conv %.12f table %.12f first-action %.12f\n", algos::NashConv(*game, uniform, /*use_state_get_policy=*/true),
              algos::NashConv(*game, uniform_table), algos::NashConv(*game, first_action));
  PrintVector("kuhn expected returns of the first-action policy:", algos::ExpectedReturns(*game->NewInitialState(), first_action, -1));
  for (int p = 0; p < 2; ++p) {
    algos::TabularBestResponse br(*game, p, &uniform_table);
    const double value = br.Value(*game->NewInitialState());
    std::unordered_map<std::string, Action> actions = br.GetBestResponseActions();
    std::printf("kuhn best response of player %d to uniform: value %.12f actions", p, value);
    for (const char* key : {"0", "1", "2", "0p", "0b", "1p", "1b", "2p", "2b", "0pb", "1pb", "2pb"}) {
      auto it = actions.find(key);
      if (it != actions.end()) std::printf(" %s=%d", key, static_cast<int>(it->second));
    }
    std::printf("\n");
  }
  algos::CFRBRSolver cfr_br(*game);
  for (int i = 1; i <= 30; ++i) {
    cfr_br.EvaluateAndUpdatePolicy();
    if (i == 1 || i == 5 || i == 30)
      std::printf("kuhn CFR-BR iter %d exploitability %.10f\n", i, algos::Exploitability(*game, *cfr_br.AveragePolicy()));
  }
  for (algos::AverageType avg : {algos::AverageType::kSimple, algos::AverageType::kFull}) {
    algos::ExternalSamplingMCCFRSolver mccfr(*game, /*seed=*/3, avg);
    std::mt19937 rng(2024);
    for (int i = 0; i < 200; ++i) mccfr.RunIteration(&rng);
    std::printf("kuhn ES-MCCFR (%s average, caller's mt19937) iter 200 nash_conv %.9f CFR-family\n",
                avg == algos::AverageType::kFull ? "full" : "simple", algos::NashConv(*game, *mccfr.AveragePolicy()));
  }
  {
    // the solver's OWN generator (seed 5), a text checkpoint half way, and the run resumed from the text
    algos::ExternalSamplingMCCFRSolver first(*game, /*seed=*/5);
    for (int i = 0; i < 60; ++i) first.RunIteration();
    const std::string text = first.Serialize();
    const size_t at = text.find("[SolverRNG]\n") + 12;
    const std::string rng_line = text.substr(at, text.find('\n', at) - at);
    unsigned long long sum = 0;
    int words = 0;
    std::istringstream in(rng_line);
    for (unsigned long long w; in >> w; ++words) sum = sum * 1000003ull + w;
    std::printf("kuhn ES-MCCFR checkpoint after 60 iterations: [SolverRNG] %d words, digest %llu\n", words, sum);
    std::unique_ptr<algos::ExternalSamplingMCCFRSolver> resumed = algos::DeserializeExternalSamplingMCCFRSolver(text);
    for (int i = 0; i < 60; ++i) resumed->RunIteration();
    std::printf("kuhn ES-MCCFR (own generator, resumed from the checkpoint) iter 120 nash_conv %.9f CFR-family\n",
                algos::NashConv(*game, *resumed->AveragePolicy()));
  }
  // MCTSBot is a Bot: x (0, 1) against o (3, 4), x to move: the only winning move is 2, and the solver proves it
  std::shared_ptr<const spiel::Game> ttt = spiel::LoadGame("tic_tac_toe");
  std::unique_ptr<spiel::State> state = ttt->NewInitialState();
  for (Action a : {0, 3, 1, 4}) state->ApplyAction(a);
  auto evaluator = std::make_shared<algos::RandomRolloutEvaluator>(2, 11);
  std::unique_ptr<spiel::Bot> bot = std::make_unique<algos::MCTSBot>(*ttt, evaluator, 2.0, 300, 50, /*solve=*/true, 5, false);
  std::printf("tic_tac_toe Bot::Step at %s -> %d\n", state->HistoryString().c_str(), static_cast<int>(bot->Step(*state)));
}

int main() {
  for (const char* g : {"tic_tac_toe", "connect_four", "hex(board_size=9)", "kuhn_poker", "leduc_poker",
                        "kuhn_poker(players=3)"})
    Playthrough(g);
  InformationStates();
  SolverAnswers();
  Cfr();
  PoliciesAndSiblingSolvers();
  std::printf("done\n");
  return 0;
}
