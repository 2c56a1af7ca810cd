// tic_tac_toe_test.cc and connect_four_test.cc also exercise the JSON struct API (TicTacToeStateStruct,
// ConnectFourStateStruct, ActionStruct: nlohmann-based, off the hot path and not offered by the mirror), so their
// sources cannot be included; what they do ON the hot path is run here with the reference's own tests/basic_tests.cc
// (compiled unmodified beside this file) and the reference's own arguments: BasicTicTacToeTests
// (tic_tac_toe_test.cc:31-35), BasicConnectFourTests (connect_four_test.cc:32-36), FastLoss (:38-58),
// BasicSerializationTest (:60-65) and ArbitrarySizeTests (:319-392).
#include <cstdio>

#include "open_spiel/spiel.h"
#include "open_spiel/tests/basic_tests.h"

namespace open_spiel {
namespace {
void BoardGameTests() {
  testing::LoadGameTest("tic_tac_toe");
  testing::NoChanceOutcomesTest(*LoadGame("tic_tac_toe"));
  testing::RandomSimTest(*LoadGame("tic_tac_toe"), 100);
  testing::RandomSimTestWithUndo(*LoadGame("tic_tac_toe"), 1);
  testing::LoadGameTest("connect_four");
  testing::NoChanceOutcomesTest(*LoadGame("connect_four"));
  testing::RandomSimTest(*LoadGame("connect_four"), 100);
  {  // FastLoss
    auto game = LoadGame("connect_four");
    auto state = game->NewInitialState();
    for (int a : {3, 3, 4, 4, 2, 2}) state->ApplyAction(a);
    SPIEL_CHECK_FALSE(state->IsTerminal());
    state->ApplyAction(1);
    SPIEL_CHECK_TRUE(state->IsTerminal());
    SPIEL_CHECK_EQ(state->Returns(), (std::vector<double>{1.0, -1.0}));
    SPIEL_CHECK_EQ(state->ToString(), ".......\n.......\n.......\n.......\n..ooo..\n.xxxx..\n");
    auto fresh = game->NewInitialState();
    SPIEL_CHECK_EQ(fresh->ToString(), game->DeserializeState(fresh->Serialize())->ToString());
  }
  {  // ArbitrarySizeTests
    auto game_4x5 = LoadGame("connect_four", {{"rows", GameParameter(4)}, {"columns", GameParameter(5)}});
    SPIEL_CHECK_EQ(game_4x5->MaxGameLength(), 20);
    testing::RandomSimTest(*game_4x5, 10);
    auto state = game_4x5->NewInitialState();
    for (int a : {0, 1, 0, 1, 0, 1}) state->ApplyAction(a);
    SPIEL_CHECK_EQ(state->ToString(), ".....\nxo...\nxo...\nxo...\n");
    state->ApplyAction(0);
    SPIEL_CHECK_TRUE(state->IsTerminal());
    SPIEL_CHECK_EQ(state->Returns(), (std::vector<double>{1.0, -1.0}));
    auto game_7x8 = LoadGame("connect_four", {{"rows", GameParameter(7)}, {"columns", GameParameter(8)}});
    SPIEL_CHECK_EQ(game_7x8->MaxGameLength(), 56);
    testing::RandomSimTest(*game_7x8, 10);
    auto state_c5 = LoadGame("connect_four", {{"x_in_row", GameParameter(5)}})->NewInitialState();
    for (int i = 0; i < 4; ++i) { state_c5->ApplyAction(0); state_c5->ApplyAction(1); }
    SPIEL_CHECK_FALSE(state_c5->IsTerminal());
    state_c5->ApplyAction(0);
    SPIEL_CHECK_TRUE(state_c5->IsTerminal());
  }
}
}  // namespace
}  // namespace open_spiel

int main() {
  open_spiel::BoardGameTests();
  std::printf("reference basic_tests on tic_tac_toe and connect_four (host mirror): passed\n");
  return 0;
}
