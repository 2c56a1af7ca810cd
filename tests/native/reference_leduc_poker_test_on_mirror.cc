// The reference's own games/leduc_poker/leduc_poker_test.cc, INCLUDED UNMODIFIED, with tests/basic_tests.cc (compiled
// unmodified beside it), against the MI355X host mirror.  BasicLeducTests() also plays 4- and 5-player leduc; the
// device packs leduc_poker for 2 and 3 players (open_spiel_amd/csrc/osg_game_poker.h), so this main() makes the same
// calls with the reference's own arguments for every configuration the engine offers: RandomSimTest (100 random
// games each: Clone, serialization round trips, legal-action masks, sorted and unique actions, observation /
// information-state tensors and strings for every player at every state, returns within bounds and summing to zero)
// on the default game, action_mapping, suit_isomorphism and 3 players; ResampleFromInfostate (100 games); the
// single_tensor observer; the always-fold / call / raise policies.  StartingPlayerTest is not called: its first
// betting action is a fold that LegalActions() does not offer (leduc_poker.cc:416-457: no fold while stakes == ante),
// applied through ApplyAction, which a release build of the reference does not check; the engine refuses every
// illegal action (osg_apply).  starting_player itself is covered against the oracle (tests/test_gpu_parity.py).
#define main reference_test_main
#include "open_spiel/games/leduc_poker/leduc_poker_test.cc"
#undef main

int main() {
  namespace testing = open_spiel::testing;
  using open_spiel::GameParameter;
  using open_spiel::LoadGame;
  testing::LoadGameTest("leduc_poker");
  testing::ChanceOutcomesTest(*LoadGame("leduc_poker"));
  testing::RandomSimTest(*LoadGame("leduc_poker"), 100);
  testing::RandomSimTest(*LoadGame("leduc_poker", {{"action_mapping", GameParameter(true)}}), 100);
  testing::RandomSimTest(*LoadGame("leduc_poker", {{"suit_isomorphism", GameParameter(true)}}), 100);
  testing::RandomSimTest(*LoadGame("leduc_poker", {{"players", GameParameter(3)}}), 100);
  testing::ResampleInfostateTest(*LoadGame("leduc_poker"), /*num_sims=*/100);
  auto observer = LoadGame("leduc_poker")->MakeObserver(open_spiel::kDefaultObsType,
                                                        open_spiel::GameParametersFromString("single_tensor"));
  testing::RandomSimTestCustomObserver(*LoadGame("leduc_poker"), observer);
  open_spiel::leduc_poker::PolicyTest();
  std::printf("reference leduc_poker_test on the host mirror: passed\n");
  return 0;
}
