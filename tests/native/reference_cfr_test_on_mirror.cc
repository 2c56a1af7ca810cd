// The reference's own algorithms/cfr_test.cc, INCLUDED UNMODIFIED, against the MI355X host mirror (the product's
// drop-in headers include/open_spiel/**; abseil includes resolve to the stand-ins under oracle/ref_shim).  Its main()
// also solves goofspiel and three matrix games through LoadGameAsTurnBased — game transforms and games outside the
// hot path — which are left out; every other test runs with the reference's own arguments: Kuhn CFR / CFR+ reach the
// Nash value -1/18 +- 1e-3 and exploitability <= 0.05, the 3- and 4-player Kuhn and 2-player leduc NashConv bounds,
// the values-table and whole-solver serialization round trips.
#define main reference_test_main
#include "open_spiel/algorithms/cfr_test.cc"
#undef main

int main() {
  algorithms::CFRTest_KuhnPoker();
  algorithms::CFRPlusTest_KuhnPoker();
  algorithms::CFRTest_KuhnPokerRunsWithThreePlayers(false, false, false);
  algorithms::CFRTest_KuhnPokerRunsWithThreePlayers(true, false, false);
  algorithms::CFRTest_KuhnPokerRunsWithThreePlayers(true, true, false);
  algorithms::CFRTest_KuhnPokerRunsWithThreePlayers(true, true, true);
  algorithms::CFRTest_GeneralMultiplePlayerTest("kuhn_poker", 3, 10, 1.0);
  algorithms::CFRTest_GeneralMultiplePlayerTest("kuhn_poker", 4, 10, 1.0);
  algorithms::CFRTest_GeneralMultiplePlayerTest("leduc_poker", 2, 10, 2.0);
  algorithms::CFRTest_InfoStateValuesTableSerialization();
  algorithms::CFRTest_CFRSolverSerialization();
  std::printf("reference cfr_test on the host mirror: 11 tests passed\n");
  return 0;
}
