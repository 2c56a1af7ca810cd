// The reference's own algorithms/outcome_sampling_mccfr_test.cc, INCLUDED UNMODIFIED, against the MI355X host
// mirror (see include/open_spiel); liars_dice (outside the hot path) is left out of main().
#define main reference_test_main
#include "open_spiel/algorithms/outcome_sampling_mccfr_test.cc"
#undef main

int main() {
  std::mt19937 rng(algorithms::kSeed);
  algorithms::MCCFR_2PGameTest("kuhn_poker", &rng, 10000, 0.17);
  algorithms::MCCFR_2PGameTest("leduc_poker", &rng, 10000, 3.07);
  algorithms::MCCFR_SerializationTest();
  std::printf("reference outcome_sampling_mccfr_test on the host mirror: 3 tests passed\n");
  return 0;
}
