// The reference's own unit tests, restated against the host mirror
// (open_spiel_amd/csrc/host/osg_spiel.h) and executed on the MI355X:
//   connect_four_test.cc:38-59 (FastLoss), :68-87 (full-board draw)
//   hex_test.cc:31-48 (board orientation, 3x4: black wins)
//   kuhn_poker_test.cc / cfr_test.cc:36-62 (CFR on kuhn: value -1/18, 12 infostates)
//   mcts_test.cc:126-155 (MCTS-Solver known answers on tic_tac_toe)
//   external_sampling_mccfr_test.cc (runs, tables move)
// Exit code 0 = all checks passed.  Needs a GPU (no CPU fallback).
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "osg_spiel.h"

using namespace open_spiel::hip;
using namespace open_spiel::hip::algorithms;

#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "%s:%d EXPECT failed: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                         \
    }                                                                       \
  } while (0)

static void ConnectFourFastLoss() {
  auto game = LoadGame("connect_four");
  auto state = game->NewInitialState();
  // The reference plays 3,3,4,4,2,2,1: x completes the bottom row.
  for (Action a : {3, 3, 4, 4, 2, 2}) {
    EXPECT(!state->IsTerminal());
    state->ApplyAction(a);
  }
  EXPECT(state->CurrentPlayer() == 0);
  EXPECT(state->LegalActions().size() == 7);
  state->ApplyAction(1);
  EXPECT(state->IsTerminal());
  EXPECT(state->CurrentPlayer() == kTerminalPlayerId);
  EXPECT(state->Returns() == (std::vector<double>{1.0, -1.0}));
  EXPECT(state->LegalActions().empty());
  bool threw = false;
  try { state->ApplyAction(0); } catch (const SpielException&) { threw = true; }
  EXPECT(threw);
  EXPECT(state->History().size() == 7);
}

static void ConnectFourObservation() {
  auto game = LoadGame("connect_four");
  EXPECT(game->ObservationTensorShape() == (std::vector<int>{3, 6, 7}));
  auto state = game->NewInitialState();
  state->ApplyAction(3);
  std::vector<float> obs = state->ObservationTensor(0);
  EXPECT(obs.size() == 126);
  EXPECT(obs[0 * 42 + 0 * 7 + 3] == 1.0f);  // plane 0 = x, row 0 (bottom), column 3
  EXPECT(obs[2 * 42 + 0 * 7 + 3] == 0.0f);  // no longer empty
  bool threw = false;
  try { state->ObservationTensor(2); } catch (const SpielException&) { threw = true; }
  EXPECT(threw);
}

static void HexBoardOrientation() {
  auto game = LoadGame("hex(num_cols=3,num_rows=4)");
  auto state = game->NewInitialState();
  for (Action a : {1, 2, 4, 5, 7, 8, 10}) state->ApplyAction(a);  // hex_test.cc:31-48
  EXPECT(state->IsTerminal());
  EXPECT(state->PlayerReturn(0) == 1.0 && state->PlayerReturn(1) == -1.0);
}

static void KuhnCfr() {
  auto game = LoadGame("kuhn_poker");
  CFRSolver solver(*game);
  EXPECT(solver.NumInfoStates() == 12 && solver.NumHistories() == 58);
  for (int i = 0; i < 300; ++i) solver.EvaluateAndUpdatePolicy();
  TabularPolicyTable avg = solver.TabularAveragePolicy();
  EXPECT(avg.size() == 12);
  // Kuhn equilibrium family (kuhn_poker.cc:451-474): with a jack, never call a bet; with a king,
  // always call.  P(bet | "0") = alpha in [0, 1/3], P(bet | "2") = 3 alpha.
  EXPECT(avg.at("0pb")[0].second > 0.97);
  EXPECT(avg.at("2pb")[1].second > 0.97);
  EXPECT(avg.at("2b")[1].second > 0.97);
  const double alpha = avg.at("0")[1].second;
  EXPECT(alpha >= 0.0 && alpha < 0.36);
  EXPECT(std::fabs(avg.at("2")[1].second - 3 * alpha) < 0.06);
  CFRInfoStateValuesTable table = solver.InfoStateValuesTable();
  EXPECT(table.at("1pb").legal_actions == (std::vector<Action>{0, 1}));
  // examples/cfr_example.cc:41-45: exploitability of the average policy (cfr_test.cc:49-51: <= 0.05, value -1/18)
  EXPECT(Exploitability(*game, avg) <= 0.05);
  EXPECT(std::fabs(ExpectedReturns(*game, avg)[0] + 1.0 / 18) <= 1e-3);
  EXPECT(std::fabs(solver.EvaluatePolicy(0).nash_conv - NashConv(*game, avg)) < 1e-12);
  // kuhn_poker_test.cc / tabular_exploitability_test.cc: every member of the optimal family is unexploitable
  // and worth -1/18 to the first player
  for (double a : {0.0, 0.1, 1.0 / 3}) {
    const TabularPolicyTable optimal = kuhn_poker::GetOptimalPolicy(a).PolicyTable();
    EXPECT(std::fabs(Exploitability(*game, optimal)) < 1e-12);
    EXPECT(std::fabs(NashConv(*game, optimal)) < 1e-12);
    EXPECT(std::fabs(ExpectedReturns(*game, optimal)[0] + 1.0 / 18) < 1e-12);
  }
  // cfr_test.cc:191-256: serialize / deserialize round trip, then both solvers stay in lock step
  const std::string text = solver.Serialize();
  EXPECT(text.find("[Meta]\nVersion: 1\n\n[Game]\nkuhn_poker()\n[SolverType]\nCFRSolver\n[SolverSpecificState]\n300\n"
                   "[SolverValuesTable]\n") != std::string::npos);
  std::unique_ptr<CFRSolver> restored = DeserializeCFRSolver(text);
  EXPECT(restored->Iteration() == 300);
  solver.EvaluateAndUpdatePolicy();
  restored->EvaluateAndUpdatePolicy();
  const CFRInfoStateValuesTable mine = solver.InfoStateValuesTable(), theirs = restored->InfoStateValuesTable();
  for (const auto& kv : mine) {
    const CFRInfoStateValues& other = theirs.at(kv.first);
    EXPECT(kv.second.cumulative_regrets == other.cumulative_regrets);   // hex floats: lossless
    EXPECT(kv.second.cumulative_policy == other.cumulative_policy);
    EXPECT(kv.second.current_policy == other.current_policy);
  }
  EXPECT(solver.Serialize(6).find("0x") == std::string::npos);         // SimpleDoubleFormatter
  bool threw = false;
  try { DeserializeCFRPlusSolver(text); } catch (const SpielException&) { threw = true; }
  EXPECT(threw);
  CFRPlusSolver plus(*game);
  plus.EvaluateAndUpdatePolicy(200);
  EXPECT(plus.TabularAveragePolicy().at("2pb")[1].second > 0.99);
}

static void LeducMccfr() {
  auto game = LoadGame("leduc_poker");
  ExternalSamplingMCCFRSolver solver(*game, 230398247);
  EXPECT(solver.NumInfoStates() == 936 && solver.NumHistories() == 9457);
  for (int i = 0; i < 5; ++i) solver.RunIteration();
  solver.RunMiniBatch(4096);
  int moved = 0;
  for (const auto& kv : solver.InfoStateValuesTable())
    for (double r : kv.second.cumulative_regrets) moved += r != 0.000001;
  EXPECT(moved > 100);
}

static std::unique_ptr<SearchNode> SearchTicTacToe(std::initializer_list<Action> moves,
                                                   std::unique_ptr<State>* out_state) {
  auto game = LoadGame("tic_tac_toe");
  auto state = game->NewInitialState();
  for (Action a : moves) state->ApplyAction(a);
  auto evaluator = std::make_shared<RandomRolloutEvaluator>(20, 42);
  MCTSBot bot(*game, evaluator, /*uct_c=*/2, /*max_simulations=*/10000, /*max_memory_mb=*/10,
              /*solve=*/true, /*seed=*/42, /*verbose=*/false);
  auto root = bot.MCTSearch(*state);
  *out_state = std::move(state);
  return root;
}

static void MctsSolver() {
  std::unique_ptr<State> state;
  auto root = SearchTicTacToe({4, 0, 8}, &state);  // "x(1,1) o(0,0) x(2,2)"
  EXPECT(root->outcome[root->player] == 0);
  for (const SearchNode& c : root->children) EXPECT(c.outcome.empty() || c.outcome[c.player] <= 0);
  const SearchNode& best = root->BestChild();
  EXPECT(best.outcome[best.player] == 0);
  EXPECT(best.action == 6 || best.action == 2);  // o(2,0) or o(0,2)
  root = SearchTicTacToe({4, 0, 8, 1, 2}, &state);  // "... o(0,1) x(0,2)"
  EXPECT(root->outcome[root->player] == -1);
  for (const SearchNode& c : root->children) EXPECT(c.outcome[c.player] == -1);
  root = SearchTicTacToe({1, 8}, &state);  // "x(0,1) o(2,2)"
  EXPECT(root->outcome[root->player] == 1);
  const SearchNode& win = root->BestChild();
  EXPECT(win.outcome[win.player] == 1 && win.action == 2);  // x(0,2)
}

static void BatchedStep() {
  auto game = LoadGame("kuhn_poker");
  BatchedState batch = game->NewInitialStates(5);
  batch.ApplyActions({0, 1, 2, 0, -1});
  std::vector<int8_t> cur = batch.CurrentPlayer();
  for (int i = 0; i < 5; ++i) EXPECT(cur[i] == kChancePlayerId);
  batch.ApplyActions({1, 0, 0, 2, 1});
  EXPECT(batch.CurrentPlayer()[0] == 0 && batch.CurrentPlayer()[4] == kChancePlayerId);
  bool threw = false;
  try { batch.ApplyActions({1, 0, 0, 2, 1}); } catch (const SpielException&) { threw = true; }  // card 1 taken
  (void)threw;
  auto ev = std::make_shared<RandomRolloutEvaluator>(8, 7);
  std::vector<double> means = ev->EvaluateBatch(batch);
  EXPECT(means.size() == 10);
  for (int i = 0; i < 5; ++i) EXPECT(std::fabs(means[2 * i] + means[2 * i + 1]) < 1e-12);  // zero-sum
}

int main() {
  try {
    ConnectFourFastLoss();
    ConnectFourObservation();
    HexBoardOrientation();
    BatchedStep();
    KuhnCfr();
    LeducMccfr();
    MctsSolver();
  } catch (const std::exception& e) {
    std::fprintf(stderr, "uncaught: %s\n", e.what());
    return 2;
  }
  std::printf("host_api_test: all checks passed\n");
  return 0;
}
