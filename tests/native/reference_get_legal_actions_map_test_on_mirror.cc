// The reference's own algorithms/get_legal_actions_map_test.cc, INCLUDED UNMODIFIED, against the MI355X host mirror.
// Its main() also loads goofspiel (outside the hot path); KuhnTest (6 + 6 infostates, two legal actions each) and
// LeducTest (936 infostates) run as they are.
#include <cstdio>
#define main reference_test_main
#include "open_spiel/algorithms/get_legal_actions_map_test.cc"
#undef main

int main() {
  KuhnTest();
  LeducTest();
  std::printf("reference get_legal_actions_map_test on the host mirror: 2 tests passed\n");
  return 0;
}
