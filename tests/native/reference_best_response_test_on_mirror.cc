// The reference's own algorithms/best_response_test.cc, INCLUDED UNMODIFIED, against the MI355X host mirror.  One of
// its 14 tests loads Kuhn poker from an .efg file (efg_game: outside the hot path) and is not called; the other 13 run
// as written: best-response actions against the uniform, optimal, first-action and exploitability-descent policies,
// and the best-response VALUE OF EVERY HISTORY (58 per responder) against the reference's golden tables —
// TabularBestResponse::Value(history) on the device's per-history pass.
#define main reference_test_main
#include "open_spiel/algorithms/best_response_test.cc"
#undef main

int main() {
  open_spiel::algorithms::KuhnPokerUniformBestResponsePid0();
  open_spiel::algorithms::KuhnPokerUniformBestResponsePid1();
  open_spiel::algorithms::KuhnPokerOptimalBestResponsePid0();
  open_spiel::algorithms::KuhnPokerOptimalBestResponsePid1();
  open_spiel::algorithms::KuhnPokerExploitabilityDescentIteration4BestResponsePid0();
  open_spiel::algorithms::KuhnPokerFirstActionBestResponsePid0();
  open_spiel::algorithms::KuhnPokerFirstActionBestResponsePid1();
  open_spiel::algorithms::KuhnPokerExploitabilityDescentMinimalSimulationPid0();
  open_spiel::algorithms::KuhnPokerUniformValueTestPid0();
  open_spiel::algorithms::KuhnPokerUniformValueTestPid1();
  open_spiel::algorithms::KuhnPokerOptimalValueTestPid0();
  open_spiel::algorithms::KuhnPokerOptimalValueTestPid1();
  open_spiel::algorithms::KuhnPokerUniformBestResponseAfterSwitchingPolicies();
  std::printf("reference best_response_test on the host mirror: 13 tests passed\n");
  return 0;
}
