// The reference's own algorithms/mcts_test.cc, INCLUDED UNMODIFIED, compiled against the MI355X host mirror
// (include/open_spiel resolves its "open_spiel/..." includes to the mirror; its abseil includes resolve to the
// private stand-ins under oracle/ref_shim) and run on the device.  Two of its ten tests load games outside the
// hot path (catch, pig) and are not called; the other eight run as written: self-play through the Bot interface
// and EvaluateBots, sampling from the prior at 0 / 1 simulations, the three MCTS-Solver known answers, the
// node budget (max_memory_mb = 1: the search collects garbage until the root is proven) and the wall-clock limit.
#define main reference_mcts_test_main
#include "open_spiel/algorithms/mcts_test.cc"
#undef main

int main() {
  open_spiel::MCTSTest_CanPlayTicTacToe();
  open_spiel::MCTSTest_CanPlayTicTacToe_LowSimulations();
  open_spiel::MCTSTest_CanPlayBothSides();
  open_spiel::MCTSTest_SolveDraw();
  open_spiel::MCTSTest_SolveLoss();
  open_spiel::MCTSTest_SolveWin();
  open_spiel::MCTSTest_GarbageCollect();
  open_spiel::MCTSTest_WallClockTimeLimit();
  std::printf("reference mcts_test on the host mirror: 8 tests passed\n");
  return 0;
}
