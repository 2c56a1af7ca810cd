// The reference's own algorithms/external_sampling_mccfr_test.cc, INCLUDED UNMODIFIED, against the MI355X host
// mirror (see include/open_spiel).  Its main() also solves liars_dice, which is outside the hot path; this main()
// calls the other tests with the reference's own arguments and ONE generator carried through them, as the
// reference does — and because RunIteration(std::mt19937*) consumes the generator exactly as the reference's does,
// the runs follow the reference's draw for draw (the NashConv values printed are the reference's own).
#define main reference_test_main
#include "open_spiel/algorithms/external_sampling_mccfr_test.cc"
#undef main

int main() {
  std::mt19937 rng(algorithms::kSeed);
  algorithms::MCCFR_2PGameTest("kuhn_poker", &rng, 1000, 0.05);
  algorithms::MCCFR_2PGameTest("leduc_poker", &rng, 1000, 2.5);
  algorithms::MCCFR_KuhnPoker3PTest(&rng);
  algorithms::MCCFR_SerializationTest();
  std::printf("reference external_sampling_mccfr_test on the host mirror: 4 tests passed\n");
  return 0;
}
