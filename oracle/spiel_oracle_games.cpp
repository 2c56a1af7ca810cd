// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see spiel_oracle.h).
// The five hot-path games, restated from the reference with the same
// data-structure shapes (cell vectors, per-call std::vector results).
#include <algorithm>
#include <array>
#include <cmath>
#include <numeric>
#include <sstream>

#include "spiel_oracle.h"

namespace osg_oracle {
namespace {

std::string JoinInts(const std::vector<int>& v, const char* sep) {
  std::string s;
  for (size_t i = 0; i < v.size(); ++i) {
    if (i) s += sep;
    s += std::to_string(v[i]);
  }
  return s;
}
// absl::StrJoin / StrCat of a double prints the shortest "%g"-like form
// ("97", "102.5").  Six significant digits is what absl's legacy double
// formatting gives (StrCat(double) == SixDigits), enough for chip counts.
std::string DoubleStr(double d) {
  char buf[64];
  snprintf(buf, sizeof(buf), "%g", d);
  return buf;
}

void CheckPlayer(const State& s, Player p) {
  if (p < 0 || p >= s.NumPlayers()) Fatal("player id out of range");
}

// ============================================================================
// tic_tac_toe  (games/tic_tac_toe/tic_tac_toe.{h,cc})
// ============================================================================
// Cell encoding = the reference enum order (tic_tac_toe.h:51-55):
// 0 empty, 1 nought (player 1), 2 cross (player 0).
constexpr int kTttEmpty = 0, kTttO = 1, kTttX = 2;

class TttGame : public Game {
 public:
  explicit TttGame(GameParams p) : Game("tic_tac_toe", std::move(p)) {}
  int NumDistinctActions() const override { return 9; }
  std::unique_ptr<State> NewInitialState() const override;
  int NumPlayers() const override { return 2; }
  double MinUtility() const override { return -1; }
  double MaxUtility() const override { return 1; }
  std::vector<int> ObservationTensorShape() const override { return {3, 3, 3}; }
  int MaxGameLength() const override { return 9; }
};

class TttState : public State {
 public:
  explicit TttState(std::shared_ptr<const Game> g) : State(std::move(g)) {
    cells_.fill(kTttEmpty);
  }
  Player CurrentPlayer() const override {  // tic_tac_toe.h:87-89
    return IsTerminal() ? kTerminalPlayerId : to_move_;
  }
  std::vector<Action> LegalActions() const override {  // tic_tac_toe.cc:138-148
    std::vector<Action> out;
    if (IsTerminal()) return out;
    for (int c = 0; c < 9; ++c)
      if (cells_[c] == kTttEmpty) out.push_back(c);
    return out;
  }
  std::string ActionToString(Player player, Action a) const override {
    // tic_tac_toe.cc:266-270
    return std::string(player == 0 ? "x" : "o") + "(" + std::to_string(a / 3) +
           "," + std::to_string(a % 3) + ")";
  }
  std::string ToString() const override {  // tic_tac_toe.cc:163-175
    std::string s;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) s += Glyph(cells_[r * 3 + c]);
      if (r < 2) s += "\n";
    }
    return s;
  }
  bool IsTerminal() const override {  // tic_tac_toe.cc:215-217
    return winner_ != kInvalidPlayer || plies_ == 9;
  }
  std::vector<double> Returns() const override {  // tic_tac_toe.cc:219-227
    if (Line(0)) return {1.0, -1.0};
    if (Line(1)) return {-1.0, 1.0};
    return {0.0, 0.0};
  }
  std::string InformationStateString(Player p) const override {
    CheckPlayer(*this, p);
    return HistoryString();
  }
  std::string ObservationString(Player p) const override {
    CheckPlayer(*this, p);
    return ToString();
  }
  void ObservationTensor(Player p, float* out, int n) const override {
    // tic_tac_toe.cc:241-251: plane index = raw cell enum value.
    CheckPlayer(*this, p);
    ORACLE_CHECK(n == 27);
    std::fill(out, out + n, 0.0f);
    for (int c = 0; c < 9; ++c) out[cells_[c] * 9 + c] = 1.0f;
  }
  using State::ObservationTensor;
  std::unique_ptr<State> Clone() const override {
    return std::unique_ptr<State>(new TttState(*this));
  }

 protected:
  void DoApplyAction(Action a) override {  // tic_tac_toe.cc:128-136
    ORACLE_CHECK(a >= 0 && a < 9 && cells_[a] == kTttEmpty);
    const Player mover = CurrentPlayer();  // PlayerToState (tic_tac_toe.cc:66-76): fatal unless 0 or 1
    if (mover != 0 && mover != 1) Fatal("Invalid player id " + std::to_string(mover));
    cells_[a] = (mover == 0) ? kTttX : kTttO;
    if (Line(to_move_)) winner_ = to_move_;
    to_move_ = 1 - to_move_;
    ++plies_;
  }

 private:
  static const char* Glyph(int v) {
    return v == kTttEmpty ? "." : (v == kTttO ? "o" : "x");
  }
  bool Line(Player p) const {  // tic_tac_toe.cc:109-120 (8 lines)
    static const int kLines[8][3] = {{0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {0, 3, 6},
                                     {1, 4, 7}, {2, 5, 8}, {0, 4, 8}, {2, 4, 6}};
    int mark = (p == 0) ? kTttX : kTttO;
    for (const auto& l : kLines)
      if (cells_[l[0]] == mark && cells_[l[1]] == mark && cells_[l[2]] == mark)
        return true;
    return false;
  }
  std::array<int, 9> cells_;
  Player to_move_ = 0;
  Player winner_ = kInvalidPlayer;
  int plies_ = 0;
};
std::unique_ptr<State> TttGame::NewInitialState() const {
  return std::unique_ptr<State>(new TttState(shared_from_this()));
}

// ============================================================================
// connect_four  (games/connect_four/connect_four.{h,cc})
// ============================================================================
// Cells: 0 empty, 1 nought (player 1, 'o'), 2 cross (player 0, 'x')
// (connect_four.h:65-69).  Board index row*cols+col, row 0 = bottom.
class C4Game : public Game {
 public:
  explicit C4Game(GameParams p) : Game("connect_four", std::move(p)) {
    // connect_four.cc:50-54,332-338
    ego_ = BoolParam("egocentric_obs_tensor", false);
    rows_ = IntParam("rows", 6);
    cols_ = IntParam("columns", 7);
    k_ = IntParam("x_in_row", 4);
  }
  int NumDistinctActions() const override { return cols_; }
  std::unique_ptr<State> NewInitialState() const override;
  int NumPlayers() const override { return 2; }
  double MinUtility() const override { return -1; }
  double MaxUtility() const override { return 1; }
  std::vector<int> ObservationTensorShape() const override {
    return {3, rows_, cols_};
  }
  int MaxGameLength() const override { return rows_ * cols_; }
  bool ego_;
  int rows_, cols_, k_;
};

class C4State : public State {
 public:
  explicit C4State(std::shared_ptr<const Game> g)
      : State(g), cfg_(static_cast<const C4Game&>(*g)) {
    grid_.assign(cfg_.rows_ * cfg_.cols_, 0);
  }
  Player CurrentPlayer() const override {  // connect_four.cc:122-128
    return IsTerminal() ? kTerminalPlayerId : to_move_;
  }
  std::vector<Action> LegalActions() const override {  // :147-156
    std::vector<Action> out;
    if (IsTerminal()) return out;
    for (int c = 0; c < cfg_.cols_; ++c)
      if (At(cfg_.rows_ - 1, c) == 0) out.push_back(c);
    return out;
  }
  std::string ActionToString(Player player, Action a) const override {
    return std::string(player == 0 ? "x" : "o") + std::to_string(a);  // :158-161
  }
  std::string ToString() const override {  // :212-222 (top row first)
    std::string s;
    for (int r = cfg_.rows_ - 1; r >= 0; --r) {
      for (int c = 0; c < cfg_.cols_; ++c) s += Glyph(At(r, c));
      s += "\n";
    }
    return s;
  }
  bool IsTerminal() const override { return outcome_ != kUnknown; }  // :277-279
  std::vector<double> Returns() const override {                      // :281-285
    if (outcome_ == 0) return {1.0, -1.0};
    if (outcome_ == 1) return {-1.0, 1.0};
    return {0.0, 0.0};
  }
  std::string InformationStateString(Player p) const override {
    CheckPlayer(*this, p);
    return HistoryString();
  }
  std::string ObservationString(Player p) const override {
    CheckPlayer(*this, p);
    return ToString();
  }
  void ObservationTensor(Player p, float* out, int n) const override {
    // :312-328.  Default planes via StateToPlayer (:75-86): x->0, o->1,
    // empty->2.  Egocentric via PlayerRelative (:299-310).
    CheckPlayer(*this, p);
    const int R = cfg_.rows_, C = cfg_.cols_;
    ORACLE_CHECK(n == 3 * R * C);
    std::fill(out, out + n, 0.0f);
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) {
        int v = At(r, c);
        int plane;
        if (cfg_.ego_) {
          if (v == 1) plane = (p == 0) ? 0 : 1;       // kNought
          else if (v == 2) plane = (p == 1) ? 0 : 1;  // kCross
          else plane = 2;
        } else {
          plane = (v == 2) ? 0 : (v == 1 ? 1 : 2);
        }
        out[(plane * R + r) * C + c] = 1.0f;
      }
  }
  using State::ObservationTensor;
  std::unique_ptr<State> Clone() const override {
    return std::unique_ptr<State>(new C4State(*this));
  }

 protected:
  void DoApplyAction(Action a) override {  // :130-145
    ORACLE_CHECK(a >= 0 && a < cfg_.cols_);
    ORACLE_CHECK(At(cfg_.rows_ - 1, a) == 0);
    int r = 0;
    while (At(r, a) != 0) ++r;
    const Player mover = CurrentPlayer();  // PlayerToState (connect_four.cc:60-69): fatal unless 0 or 1, e.g. at a terminal state
    if (mover != 0 && mover != 1) Fatal("Invalid player id " + std::to_string(mover));
    grid_[r * cfg_.cols_ + a] = (mover == 0) ? 2 : 1;
    if (AnyLine(to_move_)) {
      outcome_ = to_move_;
    } else if (Full()) {
      outcome_ = kDraw;
    }
    to_move_ = 1 - to_move_;
  }

 private:
  enum { kUnknown = 2, kDraw = 3 };  // connect_four.h:57-62
  static const char* Glyph(int v) { return v == 0 ? "." : (v == 1 ? "o" : "x"); }
  int At(int r, int c) const { return grid_[r * cfg_.cols_ + c]; }
  bool Run(int mark, int r, int c, int dr, int dc) const {  // :174-190
    const int k = cfg_.k_;
    int er = r + (k - 1) * dr, ec = c + (k - 1) * dc;
    if (er >= cfg_.rows_ || ec >= cfg_.cols_ || er < 0 || ec < 0) return false;
    for (int i = 0; i < k; ++i, r += dr, c += dc)
      if (At(r, c) != mark) return false;
    return true;
  }
  bool AnyLine(Player p) const {  // :163-172,192-201: every cell x 4 directions
    int mark = (p == 0) ? 2 : 1;
    for (int c = 0; c < cfg_.cols_; ++c)
      for (int r = 0; r < cfg_.rows_; ++r)
        if (At(r, c) == mark &&
            (Run(mark, r, c, 0, 1) || Run(mark, r, c, -1, -1) ||
             Run(mark, r, c, -1, 0) || Run(mark, r, c, -1, 1)))
          return true;
    return false;
  }
  bool Full() const {  // :203-209
    for (int c = 0; c < cfg_.cols_; ++c)
      if (At(cfg_.rows_ - 1, c) == 0) return false;
    return true;
  }
  const C4Game& cfg_;
  std::vector<int> grid_;
  Player to_move_ = 0;
  int outcome_ = kUnknown;
};
std::unique_ptr<State> C4Game::NewInitialState() const {
  return std::unique_ptr<State>(new C4State(shared_from_this()));
}

// ============================================================================
// hex  (games/hex/hex.{h,cc})
// ============================================================================
// Cell labels are the reference's CellState integers (hex.h:68-78).
enum HexLabel {
  kHexEmpty = 0,
  kHexWhite = -1, kHexWhiteEast = -2, kHexWhiteWest = -3, kHexWhiteWin = -4,
  kHexBlack = 1, kHexBlackSouth = 2, kHexBlackNorth = 3, kHexBlackWin = 4,
};

class HexGame : public Game {
 public:
  explicit HexGame(GameParams p) : Game("hex", std::move(p)) {
    // hex.cc:47-56,404-414: board_size seeds num_cols / num_rows.
    int bs = IntParam("board_size", 11);
    cols_ = IntParam("num_cols", bs);
    rows_ = IntParam("num_rows", bs);
    plain_ = BoolParam("plain_obs_tensor", false);
    std::string rep = StrParam("string_rep", "standard");
    if (rep == "standard") explicit_ = false;
    else if (rep == "explicit") explicit_ = true;
    else Fatal("Invalid string_rep " + rep);
    swap_ = BoolParam("swap", false);
  }
  int NumDistinctActions() const override { return cols_ * rows_ + (swap_ ? 1 : 0); }
  std::unique_ptr<State> NewInitialState() const override;
  int NumPlayers() const override { return 2; }
  double MinUtility() const override { return -1; }
  double MaxUtility() const override { return 1; }
  std::vector<int> ObservationTensorShape() const override {  // hex.cc:416-423
    return {plain_ ? 3 : 9, cols_, rows_};
  }
  int MaxGameLength() const override { return cols_ * rows_; }
  int cols_, rows_;
  bool plain_, explicit_, swap_;
};

class HexState : public State {
 public:
  explicit HexState(std::shared_ptr<const Game> g)
      : State(g), cfg_(static_cast<const HexGame&>(*g)) {
    cells_.assign(cfg_.cols_ * cfg_.rows_, kHexEmpty);
  }
  Player CurrentPlayer() const override {  // hex.h:96-98
    return IsTerminal() ? kTerminalPlayerId : to_move_;
  }
  std::vector<Action> LegalActions() const override {  // hex.cc:280-293
    std::vector<Action> out;
    if (IsTerminal()) return out;
    for (int i = 0; i < static_cast<int>(cells_.size()); ++i)
      if (cells_[i] == kHexEmpty) out.push_back(i);
    if (cfg_.swap_ && history_.size() == 1 && to_move_ == 1)
      out.push_back(cfg_.cols_ * cfg_.rows_);
    return out;
  }
  std::string ActionToString(Player player, Action a) const override {
    // hex.cc:295-314 ('row' there is the column letter).
    if (cfg_.swap_ && a == cfg_.cols_ * cfg_.rows_) return "swap";
    int x = static_cast<int>(a % cfg_.cols_), y = static_cast<int>(a / cfg_.cols_);
    // Always the standard form, also for string_rep=explicit: hex.cc:301 compares the value-initialised enum
    // `StringRep()` (= kStandard) instead of string_rep(), so the explicit branch (:307-310) is never taken.
    // Found by the differential run against the genuine build (tests/test_oracle_vs_reference.py).
    (void)player;
    std::string s(1, static_cast<char>('a' + x));
    return s + std::to_string(y + 1);
  }
  std::string ToString() const override {  // hex.cc:341-359
    std::string s;
    int line = 0;
    for (int i = 0; i < static_cast<int>(cells_.size()); ++i) {
      if (i && i % cfg_.cols_ == 0) {
        s += "\n";
        ++line;
        s += std::string(line, ' ');
      }
      s += Glyph(cells_[i], cfg_.explicit_);
      s += " ";
    }
    return s;
  }
  bool IsTerminal() const override { return result_ != 0; }  // :361
  std::vector<double> Returns() const override {             // :363-365
    return {result_, -result_};
  }
  std::string InformationStateString(Player p) const override {
    CheckPlayer(*this, p);
    return HistoryString();
  }
  std::string ObservationString(Player p) const override {
    CheckPlayer(*this, p);
    return ToString();
  }
  void ObservationTensor(Player p, float* out, int n) const override {
    // hex.cc:379-398
    const int N = static_cast<int>(cells_.size());
    if (cfg_.plain_) {
      ORACLE_CHECK(n == 3 * N);
      std::fill(out, out + n, 0.0f);
      for (int i = 0; i < N; ++i) {
        int plane = cells_[i] == 0 ? 2 : (cells_[i] < 0 ? 1 : 0);  // :76-93
        // view shape {3, num_cols, num_rows}, index {plane, i/cols, i%cols}
        out[(plane * cfg_.cols_ + i / cfg_.cols_) * cfg_.rows_ + i % cfg_.cols_] = 1.0f;
      }
    } else {
      CheckPlayer(*this, p);
      ORACLE_CHECK(n == 9 * N);
      std::fill(out, out + n, 0.0f);
      for (int i = 0; i < N; ++i) out[(cells_[i] + 4) * N + i] = 1.0f;
    }
  }
  using State::ObservationTensor;
  std::unique_ptr<State> Clone() const override {
    return std::unique_ptr<State>(new HexState(*this));
  }
  int LabelAt(int cell) const { return cells_[cell]; }

 protected:
  void DoApplyAction(Action a) override {  // hex.cc:229-278
    const int N = cfg_.cols_ * cfg_.rows_;
    if (cfg_.swap_ && a == N) {  // :230-244
      ORACLE_CHECK(history_.size() == 1);
      ORACLE_CHECK(to_move_ == 1);
      int first = static_cast<int>(history_[0].action);
      cells_[first] = kHexEmpty;
      int r = first / cfg_.cols_, c = first % cfg_.cols_;
      int mirrored = c * cfg_.cols_ + r;
      cells_[mirrored] = LabelFor(1, mirrored);
      to_move_ = 0;
      return;
    }
    ORACLE_CHECK(a >= 0 && a < N && cells_[a] == kHexEmpty);
    int label = LabelFor(CurrentPlayer(), static_cast<int>(a));
    cells_[a] = label;
    if (label == kHexBlackWin) {
      result_ = 1;
    } else if (label == kHexWhiteWin) {
      result_ = -1;
    } else if (label != kHexBlack && label != kHexWhite) {
      // Edge-connected, not winning: relabel the plain same-colour group
      // reachable from the new stone (:252-276).
      int plain = (to_move_ == 0) ? kHexBlack : kHexWhite;
      std::vector<int> stack = {static_cast<int>(a)};
      while (!stack.empty()) {
        int cur = stack.back();
        stack.pop_back();
        for (int nb : Neighbours(cur)) {
          if (cells_[nb] == plain) {
            cells_[nb] = label;
            stack.push_back(nb);
          }
        }
      }
    }
    to_move_ = 1 - to_move_;
  }

 private:
  static const char* Glyph(int label, bool explicit_rep) {  // hex.cc:173-227
    if (!explicit_rep) return label == 0 ? "." : (label < 0 ? "o" : "x");
    switch (label) {
      case kHexEmpty: return ".";
      case kHexWhite: return "o";
      case kHexWhiteWin: return "O";
      case kHexWhiteWest: return "p";
      case kHexWhiteEast: return "q";
      case kHexBlack: return "x";
      case kHexBlackWin: return "X";
      case kHexBlackNorth: return "y";
      case kHexBlackSouth: return "z";
    }
    Fatal("Unknown hex label");
  }
  std::vector<int> Neighbours(int cell) const {  // hex.cc:316-329
    const int C = cfg_.cols_, N = static_cast<int>(cells_.size());
    bool north = cell < C, south = cell >= N - C;
    bool west = cell % C == 0, east = cell % C == C - 1;
    std::vector<int> out;
    if (!north) out.push_back(cell - C);
    if (!north && !east) out.push_back(cell - C + 1);
    if (!east) out.push_back(cell + 1);
    if (!south) out.push_back(cell + C);
    if (!south && !west) out.push_back(cell + C - 1);
    if (!west) out.push_back(cell - 1);
    return out;
  }
  int LabelFor(Player player, int move) const {  // hex.cc:108-171
    const int C = cfg_.cols_, N = static_cast<int>(cells_.size());
    bool a = false, b = false;  // black: north/south; white: west/east
    if (player == 0) {
      if (move < C) a = true;
      else if (move >= N - C) b = true;
      for (int nb : Neighbours(move)) {
        if (cells_[nb] == kHexBlackNorth) a = true;
        else if (cells_[nb] == kHexBlackSouth) b = true;
      }
      if (a && b) return kHexBlackWin;
      if (a) return kHexBlackNorth;
      if (b) return kHexBlackSouth;
      return kHexBlack;
    }
    ORACLE_CHECK(player == 1);
    if (move % C == 0) a = true;
    else if (move % C == C - 1) b = true;
    for (int nb : Neighbours(move)) {
      if (cells_[nb] == kHexWhiteWest) a = true;
      else if (cells_[nb] == kHexWhiteEast) b = true;
    }
    if (a && b) return kHexWhiteWin;
    if (a) return kHexWhiteWest;
    if (b) return kHexWhiteEast;
    return kHexWhite;
  }
  const HexGame& cfg_;
  std::vector<int> cells_;
  Player to_move_ = 0;
  double result_ = 0;  // black's perspective
};
std::unique_ptr<State> HexGame::NewInitialState() const {
  return std::unique_ptr<State>(new HexState(shared_from_this()));
}

// ============================================================================
// kuhn_poker  (games/kuhn_poker/kuhn_poker.{h,cc})
// ============================================================================
class KuhnGame : public Game {
 public:
  explicit KuhnGame(GameParams p) : Game("kuhn_poker", std::move(p)) {
    n_ = IntParam("players", 2);
    if (n_ < 2 || n_ > 10) Fatal("kuhn_poker: players must be in [2,10]");
  }
  int NumDistinctActions() const override { return 2; }
  std::unique_ptr<State> NewInitialState() const override;
  int MaxChanceOutcomes() const override { return n_ + 1; }
  int NumPlayers() const override { return n_; }
  double MinUtility() const override { return -2; }
  double MaxUtility() const override { return (n_ - 1) * 2; }
  std::vector<int> InformationStateTensorShape() const override {
    return {6 * n_ - 1};  // kuhn_poker.cc:395-403
  }
  std::vector<int> ObservationTensorShape() const override {
    return {3 * n_ + 1};  // :405-410
  }
  int MaxGameLength() const override { return n_ * 2 - 1; }
  int MaxChanceNodesInHistory() const override { return n_; }
  bool HasChance() const override { return true; }
  int n_;
};

class KuhnState : public State {
 public:
  explicit KuhnState(std::shared_ptr<const Game> g)
      : State(g),
        holder_(g->NumPlayers() + 1, kInvalidPlayer),
        pot_(g->NumPlayers()),
        contrib_(g->NumPlayers(), 1) {}
  Player CurrentPlayer() const override {  // kuhn_poker.cc:181-188
    if (IsTerminal()) return kTerminalPlayerId;
    int h = static_cast<int>(history_.size());
    return h < num_players_ ? kChancePlayerId : h % num_players_;
  }
  std::vector<Action> LegalActions() const override {  // :231-242
    if (IsTerminal()) return {};
    if (IsChanceNode()) {
      std::vector<Action> out;
      for (int c = 0; c < static_cast<int>(holder_.size()); ++c)
        if (holder_[c] == kInvalidPlayer) out.push_back(c);
      return out;
    }
    return {0, 1};
  }
  ActionsAndProbs ChanceOutcomes() const override {  // :329-337
    ORACLE_CHECK(IsChanceNode());
    ActionsAndProbs out;
    double p = 1.0 / (num_players_ + 1 - static_cast<int>(history_.size()));
    for (int c = 0; c < static_cast<int>(holder_.size()); ++c)
      if (holder_[c] == kInvalidPlayer) out.push_back({c, p});
    return out;
  }
  std::string ActionToString(Player player, Action a) const override {  // :244-251
    if (player == kChancePlayerId) return "Deal:" + std::to_string(a);
    return a == 0 ? "Pass" : "Bet";
  }
  std::string ToString() const override {  // :253-268
    std::string s;
    int h = static_cast<int>(history_.size());
    for (int i = 0; i < h && i < num_players_; ++i) {
      if (!s.empty()) s += ' ';
      s += std::to_string(history_[i].action);
    }
    if (h > num_players_) s += ' ';
    for (int i = num_players_; i < h; ++i) s += history_[i].action ? 'b' : 'p';
    return s;
  }
  bool IsTerminal() const override { return winner_ != kInvalidPlayer; }  // :270
  std::vector<double> Returns() const override {                           // :272-283
    std::vector<double> r(num_players_, 0.0);
    if (!IsTerminal()) return r;
    for (Player p = 0; p < num_players_; ++p) {
      int paid = Bet(p) ? 2 : 1;
      r[p] = (p == winner_) ? (pot_ - paid) : -paid;
    }
    return r;
  }
  // Observer strings, kuhn_poker.cc:109-166 with kInfoStateObsType /
  // kDefaultObsType (observer.h:288-298).
  std::string InformationStateString(Player p) const override {
    CheckPlayer(*this, p);
    std::string s;
    int h = static_cast<int>(history_.size());
    if (h > p) s += std::to_string(history_[p].action);
    for (int i = num_players_; i < h; ++i) s += history_[i].action ? 'b' : 'p';
    return s;
  }
  std::string ObservationString(Player p) const override {
    CheckPlayer(*this, p);
    std::string s;
    int h = static_cast<int>(history_.size());
    if (h > p) {
      s += std::to_string(history_[p].action);
      for (Player q = 0; q < num_players_; ++q) s += std::to_string(contrib_[q]);
    }
    return s;
  }
  // KuhnObserver::WriteTensor, kuhn_poker.cc:72-107.
  void InformationStateTensor(Player p, float* out, int n) const override {
    CheckPlayer(*this, p);
    const int P = num_players_;
    ORACLE_CHECK(n == 6 * P - 1);
    std::fill(out, out + n, 0.0f);
    out[p] = 1;
    int h = static_cast<int>(history_.size());
    if (h > p) out[P + history_[p].action] = 1;
    float* bet = out + P + (P + 1);
    for (int i = P; i < h; ++i) bet[(i - P) * 2 + history_[i].action] = 1;
  }
  void ObservationTensor(Player p, float* out, int n) const override {
    CheckPlayer(*this, p);
    const int P = num_players_;
    ORACLE_CHECK(n == 3 * P + 1);
    std::fill(out, out + n, 0.0f);
    out[p] = 1;
    int h = static_cast<int>(history_.size());
    if (h > p) out[P + history_[p].action] = 1;
    float* pot = out + P + (P + 1);
    for (Player q = 0; q < P; ++q) pot[q] = static_cast<float>(contrib_[q]);
  }
  using State::InformationStateTensor;
  using State::ObservationTensor;
  std::unique_ptr<State> Clone() const override {
    return std::unique_ptr<State>(new KuhnState(*this));
  }

 protected:
  void DoApplyAction(Action a) override {  // kuhn_poker.cc:190-229
    const int P = num_players_;
    int h = static_cast<int>(history_.size());
    if (h < P) {
      ORACLE_CHECK(a >= 0 && a <= P && holder_[a] == kInvalidPlayer);
      holder_[a] = h;
    } else if (a == 1) {
      if (first_bettor_ == kInvalidPlayer) first_bettor_ = CurrentPlayer();
      pot_ += 1;
      contrib_[CurrentPlayer()] += 1;
    } else {
      ORACLE_CHECK(a == 0);
    }
    history_.push_back({CurrentPlayer(), a});  // temporarily, for Bet()
    int acted = static_cast<int>(history_.size()) - P;
    if (first_bettor_ == kInvalidPlayer && acted == P) {
      winner_ = holder_[P];
      if (winner_ == kInvalidPlayer) winner_ = holder_[P - 1];
    } else if (first_bettor_ != kInvalidPlayer && acted == P + first_bettor_) {
      for (int card = P; card >= 0; --card) {
        Player q = holder_[card];
        if (q != kInvalidPlayer && Bet(q)) {
          winner_ = q;
          break;
        }
      }
      ORACLE_CHECK(winner_ != kInvalidPlayer);
    }
    history_.pop_back();
  }

 private:
  bool Bet(Player p) const {  // DidBet, kuhn_poker.cc:339-349
    if (first_bettor_ == kInvalidPlayer) return false;
    if (p == first_bettor_) return true;
    if (p > first_bettor_) return history_[num_players_ + p].action == 1;
    return history_[num_players_ * 2 + p].action == 1;
  }
  int first_bettor_ = kInvalidPlayer;
  std::vector<int> holder_;  // card -> player
  int winner_ = kInvalidPlayer;
  int pot_;
  std::vector<int> contrib_;
};
std::unique_ptr<State> KuhnGame::NewInitialState() const {
  return std::unique_ptr<State>(new KuhnState(shared_from_this()));
}

// ============================================================================
// leduc_poker  (games/leduc_poker/leduc_poker.{h,cc})
// ============================================================================
constexpr int kNoCard = -10000;  // leduc_poker.h:60
class LeducGame : public Game {
 public:
  explicit LeducGame(GameParams p) : Game("leduc_poker", std::move(p)) {
    // leduc_poker.cc:55-59,742-751
    n_ = IntParam("players", 2);
    mapping_ = BoolParam("action_mapping", false);
    iso_ = BoolParam("suit_isomorphism", false);
    starter_ = IntParam("starting_player", 0);
    if (n_ < 2 || n_ > 10) Fatal("leduc_poker: players must be in [2,10]");
    cards_ = (n_ + 1) * 2;
  }
  int NumDistinctActions() const override { return 3; }
  std::unique_ptr<State> NewInitialState() const override;
  int MaxChanceOutcomes() const override { return iso_ ? cards_ / 2 : cards_; }
  int NumPlayers() const override { return n_; }
  double MinUtility() const override { return -13; }           // :853-861
  double MaxUtility() const override { return (n_ - 1) * 13; }  // :841-851
  int MaxBetsPerRound() const { return 3 * n_ - 2; }
  int MaxGameLength() const override { return 2 * MaxBetsPerRound(); }
  int MaxChanceNodesInHistory() const override { return 3; }
  bool HasChance() const override { return true; }
  std::vector<int> InformationStateTensorShape() const override {  // :811-820
    return {n_ + (iso_ ? cards_ : cards_ * 2) + MaxGameLength() * 2};
  }
  std::vector<int> ObservationTensorShape() const override {  // :822-831
    return {n_ + (iso_ ? cards_ : cards_ * 2) + n_};
  }
  int n_, cards_, starter_;
  bool mapping_, iso_;
};

class LeducState : public State {
 public:
  explicit LeducState(std::shared_ptr<const Game> g)
      : State(g), cfg_(static_cast<const LeducGame&>(*g)) {
    // leduc_poker.cc:241-286
    const int P = cfg_.n_;
    pot_ = P;
    cards_left_ = cfg_.cards_;
    live_ = P;
    is_winner_.assign(P, false);
    hole_.assign(P, kNoCard);
    chips_.assign(P, 99.0);
    paid_.assign(P, 1);
    out_.assign(P, false);
    deck_.resize(cfg_.cards_);
    std::iota(deck_.begin(), deck_.end(), 0);
  }
  Player CurrentPlayer() const override {  // :288-294
    return IsTerminal() ? kTerminalPlayerId : actor_;
  }
  std::vector<Action> LegalActions() const override {  // :416-457
    if (IsTerminal()) return {};
    std::vector<Action> out;
    if (actor_ == kChancePlayerId) {
      const int D = static_cast<int>(deck_.size());
      if (cfg_.iso_) {
        for (int c = 0; c < D / 2; ++c)
          if (deck_[2 * c] != kNoCard || deck_[2 * c + 1] != kNoCard)
            out.push_back(c);
      } else {
        for (int c = 0; c < D; ++c)
          if (deck_[c] != kNoCard) out.push_back(c);
      }
      return out;
    }
    if (cfg_.mapping_) return {0, 1, 2};
    if (stakes_ > paid_[actor_]) out.push_back(0);
    out.push_back(1);
    if (raises_ < 2) out.push_back(2);
    return out;
  }
  ActionsAndProbs ChanceOutcomes() const override {  // :546-571
    ORACLE_CHECK(IsChanceNode());
    ActionsAndProbs out;
    const int D = static_cast<int>(deck_.size());
    const double p = 1.0 / cards_left_;
    if (cfg_.iso_) {
      for (int c = 0; c < D / 2; ++c) {
        bool a = deck_[2 * c] != kNoCard, b = deck_[2 * c + 1] != kNoCard;
        if (a && b) out.push_back({c, p * 2});
        else if (a || b) out.push_back({c, p});
      }
      return out;
    }
    for (int c = 0; c < D; ++c)
      if (deck_[c] != kNoCard) out.push_back({c, p});
    return out;
  }
  std::string ActionToString(Player player, Action a) const override {
    // :459-461,869-875
    if (player == kChancePlayerId) return "Chance outcome:" + std::to_string(a);
    return ActionName(a);
  }
  std::string ToString() const override {  // :463-496
    const int P = num_players_;
    std::string s = "Round: " + std::to_string(round_) +
                    "\nPlayer: " + std::to_string(actor_) +
                    "\nPot: " + std::to_string(pot_) +
                    "\nMoney (player_0 player_1" + (P > 2 ? " [...]):" : "):");
    for (Player p = 0; p < P; ++p) s += " " + DoubleStr(chips_[p]);
    s += std::string("\nCards (public player_0 player_1") +
         (P > 2 ? " [...]): " : "): ") + std::to_string(board_) + " ";
    for (Player p = 0; p < P; ++p) s += std::to_string(hole_[p]) + " ";
    s += "\nRound 1 sequence: ";
    for (size_t i = 0; i < seq1_.size(); ++i) {
      if (i) s += ", ";
      s += ActionName(seq1_[i]);
    }
    s += "\nRound 2 sequence: ";
    for (size_t i = 0; i < seq2_.size(); ++i) {
      if (i) s += ", ";
      s += ActionName(seq2_[i]);
    }
    s += "\n";
    return s;
  }
  bool IsTerminal() const override {  // :498-500
    return live_ == 1 || (round_ == 2 && RoundOver());
  }
  std::vector<double> Returns() const override {  // :502-514
    std::vector<double> r(num_players_, 0.0);
    if (!IsTerminal()) return r;
    for (Player p = 0; p < num_players_; ++p) r[p] = chips_[p] - 100;
    return r;
  }
  // LeducObserver::StringFrom, leduc_poker.cc:198-239.
  std::string InformationStateString(Player p) const override {
    return ObsString(p, /*perfect_recall=*/true);
  }
  std::string ObservationString(Player p) const override {
    return ObsString(p, /*perfect_recall=*/false);
  }
  // LeducObserver::WriteTensor, leduc_poker.cc:103-192.
  void InformationStateTensor(Player p, float* out, int n) const override {
    WriteTensor(p, true, out, n);
  }
  void ObservationTensor(Player p, float* out, int n) const override {
    WriteTensor(p, false, out, n);
  }
  using State::InformationStateTensor;
  using State::ObservationTensor;
  std::unique_ptr<State> Clone() const override {
    return std::unique_ptr<State>(new LeducState(*this));
  }
  // Extra getters used by tests (leduc_poker.h:96-125).
  int round() const { return round_; }
  int pot() const { return pot_; }
  int public_card() const { return board_; }
  int private_card(Player p) const { return hole_[p]; }

 protected:
  void DoApplyAction(Action a) override {  // leduc_poker.cc:298-414
    if (actor_ == kChancePlayerId) {
      ORACLE_CHECK(a >= 0 && a < static_cast<Action>(deck_.size()));
      if (cfg_.iso_) {
        ORACLE_CHECK(deck_[a * 2] != kNoCard || deck_[a * 2 + 1] != kNoCard);
      } else {
        ORACLE_CHECK(deck_[a] != kNoCard);
      }
      if (dealt_ < num_players_) {
        DealHole(dealt_, static_cast<int>(a));
      } else {
        board_ = TakeCard(static_cast<int>(a));
        --cards_left_;
        actor_ = NextActor();
      }
      return;
    }
    if (cfg_.mapping_) {  // :333-345
      if (a == 0 && stakes_ <= paid_[actor_]) a = 1;
      else if (a == 2 && raises_ >= 2) a = 1;
    }
    if (a == 0) {  // fold
      Record(0);
      out_[actor_] = true;
      --live_;
      Advance(/*may_start_round=*/true);
    } else if (a == 1) {  // call / check
      ORACLE_CHECK(stakes_ >= paid_[actor_]);
      Pay(actor_, stakes_ - paid_[actor_]);
      ++calls_;
      Record(1);
      Advance(true);
    } else if (a == 2) {  // raise
      ORACLE_CHECK(raises_ < 2);
      int to_call = stakes_ - paid_[actor_];
      ORACLE_CHECK(to_call >= 0);
      if (to_call > 0) Pay(actor_, to_call);
      int bump = (round_ == 1) ? 2 : 4;  // leduc_poker.h:65-66
      stakes_ += bump;
      Pay(actor_, bump);
      ++raises_;
      calls_ = 0;
      Record(2);
      Advance(false);
    } else {
      Fatal("leduc: invalid move " + std::to_string(a));
    }
  }

 private:
  static std::string ActionName(Action a) {
    if (a == 0) return "Fold";
    if (a == 1) return "Call";
    if (a == 2) return "Raise";
    Fatal("Unknown action");
  }
  int ObservableCards() const {
    return cfg_.iso_ ? static_cast<int>(deck_.size()) / 2
                     : static_cast<int>(deck_.size());
  }
  int TakeCard(int move) {  // returns the card value recorded for `move`
    if (cfg_.iso_) {
      if (deck_[move * 2] != kNoCard) deck_[move * 2] = kNoCard;
      else if (deck_[move * 2 + 1] != kNoCard) deck_[move * 2 + 1] = kNoCard;
      else Fatal("Suit isomorphism error.");
      return move;
    }
    int card = deck_[move];
    deck_[move] = kNoCard;
    return card;
  }
  void DealHole(Player p, int move) {  // SetPrivate, :706-727
    hole_[p] = TakeCard(move);
    --cards_left_;
    ++dealt_;
    if (dealt_ == num_players_) actor_ = cfg_.starter_;
  }
  void Pay(Player p, int amount) {  // Ante, :700-704
    pot_ += amount;
    paid_[p] += amount;
    chips_[p] -= amount;
  }
  void Record(int move) {  // SequenceAppendMove, :691-698
    (round_ == 1 ? seq1_ : seq2_).push_back(move);
  }
  bool RoundOver() const {  // ReadyForNextRound, :680-683
    return (raises_ == 0 && calls_ == live_) ||
           (raises_ > 0 && calls_ == live_ - 1);
  }
  void Advance(bool may_start_round) {
    if (IsTerminal()) {
      Showdown();
    } else if (may_start_round && RoundOver()) {
      ORACLE_CHECK(round_ == 1);  // NewRound, :685-691
      ++round_;
      raises_ = 0;
      calls_ = 0;
      actor_ = kChancePlayerId;
    } else {
      actor_ = NextActor();
    }
  }
  Player NextActor() const {  // NextPlayer, :573-591
    const int P = num_players_;
    int from = (actor_ == kChancePlayerId) ? (cfg_.starter_ + P - 1) % P : actor_;
    for (int i = 1; i <= P; ++i) {
      Player q = (from + i) % P;
      if (!out_[q]) return q;
    }
    Fatal("leduc: no next player");
  }
  int HandRank(Player p) const {  // RankHand, :593-626
    int lo = board_, hi = hole_[p];
    if (lo > hi) std::swap(lo, hi);
    if (cfg_.iso_) {
      int n = static_cast<int>(deck_.size()) / 2;
      return lo == hi ? n * n + lo : hi * n + lo;
    }
    int n = static_cast<int>(deck_.size());
    if (lo % 2 == 0 && hi == lo + 1) return n * n + lo;
    return (hi / 2) * n + (lo / 2);
  }
  void Showdown() {  // ResolveWinner, :628-678
    const int P = num_players_;
    if (live_ == 1) {
      for (Player p = 0; p < P; ++p)
        if (!out_[p]) {
          winners_ = 1;
          is_winner_[p] = true;
          chips_[p] += pot_;
          pot_ = 0;
          return;
        }
      return;
    }
    ORACLE_CHECK(board_ != kNoCard);
    int best = -1;
    winners_ = 0;
    std::fill(is_winner_.begin(), is_winner_.end(), false);
    for (Player p = 0; p < P; ++p) {
      if (out_[p]) continue;
      int rank = HandRank(p);
      if (rank > best) {
        best = rank;
        std::fill(is_winner_.begin(), is_winner_.end(), false);
        is_winner_[p] = true;
        winners_ = 1;
      } else if (rank == best) {
        is_winner_[p] = true;
        ++winners_;
      }
    }
    ORACLE_CHECK(winners_ >= 1 && winners_ <= P);
    for (Player p = 0; p < P; ++p)
      if (is_winner_[p]) chips_[p] += static_cast<double>(pot_) / winners_;
    pot_ = 0;
  }
  std::string ObsString(Player p, bool perfect_recall) const {
    CheckPlayer(*this, p);
    std::string s = "[Observer: " + std::to_string(p) + "][Private: " +
                    std::to_string(hole_[p]) + "]";
    s += "[Round " + std::to_string(round_) + "][Player: " +
         std::to_string(actor_) + "][Pot: " + std::to_string(pot_) + "][Money: ";
    for (Player q = 0; q < num_players_; ++q) {
      if (q) s += " ";
      s += DoubleStr(chips_[q]);
    }
    s += "]";
    if (board_ != kNoCard) s += "[Public: " + std::to_string(board_) + "]";
    if (perfect_recall) {
      s += "[Round1: " + JoinInts(seq1_, " ") + "][Round2: " +
           JoinInts(seq2_, " ") + "]";
    } else {
      s += "[Ante: " + JoinInts(paid_, " ") + "]";
    }
    return s;
  }
  void WriteTensor(Player p, bool perfect_recall, float* out, int n) const {
    CheckPlayer(*this, p);
    const int P = num_players_, K = ObservableCards();
    const int bets = 3 * P - 2;
    ORACLE_CHECK(n == P + 2 * K + (perfect_recall ? 2 * bets * 2 : P));
    std::fill(out, out + n, 0.0f);
    out[p] = 1;                                      // player[P]
    if (hole_[p] != kNoCard) out[P + hole_[p]] = 1;  // private_card[K]
    if (board_ != kNoCard) out[P + K + board_] = 1;  // community_card[K]
    float* tail = out + P + 2 * K;
    if (perfect_recall) {  // betting[2, bets, 2]: call->[1,0], raise->[0,1]
      for (int r = 0; r < 2; ++r) {
        const auto& seq = r == 0 ? seq1_ : seq2_;
        for (size_t i = 0; i < seq.size(); ++i) {
          if (seq[i] == 1) tail[(r * bets + i) * 2 + 0] = 1;
          else if (seq[i] == 2) tail[(r * bets + i) * 2 + 1] = 1;
        }
      }
    } else {  // pot_contribution[P]
      for (Player q = 0; q < P; ++q) tail[q] = static_cast<float>(paid_[q]);
    }
  }

  const LeducGame& cfg_;
  Player actor_ = kChancePlayerId;
  int calls_ = 0, raises_ = 0, round_ = 1, stakes_ = 1, winners_ = -1;
  int pot_, board_ = kNoCard, cards_left_, dealt_ = 0, live_;
  std::vector<bool> is_winner_;
  std::vector<int> hole_;
  std::vector<int> deck_;
  std::vector<double> chips_;
  std::vector<int> paid_;
  std::vector<bool> out_;
  std::vector<int> seq1_, seq2_;
};
std::unique_ptr<State> LeducGame::NewInitialState() const {
  return std::unique_ptr<State>(new LeducState(shared_from_this()));
}

void RejectUnknownParams(const GameParams& given,
                         std::initializer_list<const char*> known) {
  for (const auto& kv : given) {
    if (kv.first == "name") continue;
    bool ok = false;
    for (const char* k : known) ok = ok || kv.first == k;
    if (!ok) Fatal("Unknown parameter '" + kv.first + "'");  // spiel.cc:65-90
  }
}

}  // namespace

std::shared_ptr<const Game> LoadGame(const std::string& game_string) {
  GameParams params = ParseGameString(game_string);
  auto it = params.find("name");
  if (it == params.end()) Fatal("No game name in '" + game_string + "'");
  const std::string name = it->second.s;
  params.erase("name");
  if (name == "tic_tac_toe") {
    RejectUnknownParams(params, {});
    return std::make_shared<TttGame>(params);
  }
  if (name == "connect_four") {
    RejectUnknownParams(params, {"egocentric_obs_tensor", "rows", "columns", "x_in_row"});
    return std::make_shared<C4Game>(params);
  }
  if (name == "hex") {
    RejectUnknownParams(params, {"board_size", "num_cols", "num_rows",
                                 "plain_obs_tensor", "string_rep", "swap"});
    return std::make_shared<HexGame>(params);
  }
  if (name == "kuhn_poker") {
    RejectUnknownParams(params, {"players"});
    return std::make_shared<KuhnGame>(params);
  }
  if (name == "leduc_poker") {
    RejectUnknownParams(params, {"players", "action_mapping", "suit_isomorphism",
                                 "starting_player"});
    return std::make_shared<LeducGame>(params);
  }
  Fatal("Unknown game '" + name + "'");
}

}  // namespace osg_oracle
