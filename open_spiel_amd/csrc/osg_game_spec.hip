// Host side: game strings -> validated GameSpec (description + device Params).
// Restates the parameter handling of open_spiel/game_parameters.cc:172-227 and
// the per-game parameter_specification blocks (connect_four.cc:50-54,
// hex.cc:47-56, kuhn_poker.cc:36-57, leduc_poker.cc:55-59).
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "osg_internal.h"

namespace osg {

namespace {
thread_local std::string g_last_error;

struct Value {
  enum Kind { kBool, kInt, kDouble, kString } kind = kString;
  bool b = false;
  int i = 0;
  std::string s;
  std::string str() const {
    if (kind == kBool) return b ? "True" : "False";
    if (kind == kInt) return std::to_string(i);
    return s;
  }
};

Value parse_value(const std::string& s) {  // game_parameters.cc:172-198
  Value v;
  v.s = s;
  if (s == "True" || s == "true") { v.kind = Value::kBool; v.b = true; }
  else if (s == "False" || s == "false") { v.kind = Value::kBool; v.b = false; }
  else if (!s.empty() && s.find_first_not_of("+-0123456789") == std::string::npos) {
    v.kind = Value::kInt;
    try { v.i = std::stoi(s); } catch (...) { v.kind = Value::kString; }
  } else if (!s.empty() && s.find_first_not_of("+-0123456789.") == std::string::npos) {
    v.kind = Value::kDouble;
  }
  return v;
}

bool split_game_string(const std::string& gs, std::string* name,
                       std::map<std::string, Value>* kv, std::string* err) {
  size_t open = gs.find('(');
  if (open == std::string::npos) { *name = gs; return true; }
  *name = gs.substr(0, open);
  int depth = 1;
  size_t start = open + 1;
  long eq = -1;
  for (size_t i = start; i < gs.size(); ++i) {
    char c = gs[i];
    if (c == '(') ++depth;
    if (c == ')') --depth;
    if (c == '=' && depth == 1) eq = static_cast<long>(i);
    bool sep = (c == ',' && depth == 1), end = (c == ')' && depth == 0 && i > start + 1);
    if (sep || end) {
      if (eq < 0) { *err = "malformed game string '" + gs + "'"; return false; }
      (*kv)[gs.substr(start, eq - start)] = parse_value(gs.substr(eq + 1, i - eq - 1));
      start = i + 1;
      eq = -1;
    }
  }
  if (depth > 0) { *err = "Missing closing bracket ')'."; return false; }
  return true;
}

struct ParamReader {
  std::map<std::string, Value> given;
  std::map<std::string, std::string>* all;
  std::string err;
  std::map<std::string, bool> used;
  int get_int(const char* k, int def) {
    auto it = given.find(k);
    int v = def;
    if (it != given.end()) {
      used[k] = true;
      if (it->second.kind != Value::kInt) { err = std::string("Wrong type for parameter ") + k; return def; }
      v = it->second.i;
    }
    (*all)[k] = std::to_string(v);
    return v;
  }
  bool get_bool(const char* k, bool def) {
    auto it = given.find(k);
    bool v = def;
    if (it != given.end()) {
      used[k] = true;
      if (it->second.kind != Value::kBool) { err = std::string("Wrong type for parameter ") + k; return def; }
      v = it->second.b;
    }
    (*all)[k] = v ? "True" : "False";
    return v;
  }
  std::string get_str(const char* k, const std::string& def) {
    auto it = given.find(k);
    std::string v = def;
    if (it != given.end()) {
      used[k] = true;
      if (it->second.kind != Value::kString) { err = std::string("Wrong type for parameter ") + k; return def; }
      v = it->second.s;
    }
    (*all)[k] = v;
    return v;
  }
  bool finish() {  // unknown keys are fatal, spiel.cc:65-90
    for (const auto& kv : given)
      if (!used.count(kv.first)) { err = "Unknown parameter '" + kv.first + "'"; return false; }
    return err.empty();
  }
};

template <int NW>
void fill_hex(typename HexT<NW>::Params* p, int cols, int rows, bool swap, bool plain) {
  using B = typename HexT<NW>::Bits;
  auto zero = [](B* b) { for (int i = 0; i < NW; ++i) b->w[i] = 0; };
  auto set = [](B* b, int c) { if ((c >> 5) < NW) b->w[c >> 5] |= 1u << (c & 31); };
  p->words = 4 * NW + 1;
  p->cols = cols; p->rows = rows; p->cells = cols * rows; p->swap = swap; p->plain_obs = plain;
  zero(&p->board); zero(&p->col_first); zero(&p->col_last); zero(&p->row_first); zero(&p->row_last);
  for (int c = 0; c < cols * rows; ++c) {
    set(&p->board, c);
    if (c % cols == 0) set(&p->col_first, c);
    if (c % cols == cols - 1) set(&p->col_last, c);
    if (c < cols) set(&p->row_first, c);
    if (c >= cols * rows - cols) set(&p->row_last, c);
  }
}
}  // namespace

int set_error(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
const std::string& last_error_string() { return g_last_error; }

int parse_game(const char* game_string, GameSpec* out) {
  if (!game_string || !out) return set_error(OSG_ERR_INVALID, "null argument");
  std::string name, err;
  ParamReader rd;
  rd.all = &out->params;
  out->params.clear();
  if (!split_game_string(game_string, &name, &rd.given, &err)) return set_error(OSG_ERR_INVALID, err);
  osg_game_desc& d = out->desc;
  memset(&d, 0, sizeof(d));
  d.num_players = 2;
  d.min_utility = -1;
  d.max_utility = 1;

  if (name == "tic_tac_toe") {
    if (!rd.finish()) return set_error(OSG_ERR_INVALID, rd.err);
    d.game_kind = kTtt;
    d.num_distinct_actions = 9;
    d.max_game_length = 9;
    d.obs_size = 27; d.obs_rank = 3; d.obs_shape[0] = 3; d.obs_shape[1] = 3; d.obs_shape[2] = 3;
    d.state_words = 1; d.state_word_bytes = 4;
    out->ttt.words = 1;
  } else if (name == "connect_four") {
    bool ego = rd.get_bool("egocentric_obs_tensor", false);
    int rows = rd.get_int("rows", 6), cols = rd.get_int("columns", 7), k = rd.get_int("x_in_row", 4);
    if (!rd.finish()) return set_error(OSG_ERR_INVALID, rd.err);
    if (rows < 1 || cols < 1 || k < 1) return set_error(OSG_ERR_INVALID, "connect_four: bad dimensions");
    // two u64 planes while (rows + 1) * columns <= 64, four (two words per colour) up to 128 bits — 8 x 8 ... 10 x 10,
    // 7 x 15 ...; the reference has no upper bound (connect_four.cc:50-54): larger boards stay unsupported here
    if ((rows + 1) * cols > 128 || cols > 32)
      return set_error(OSG_ERR_UNSUPPORTED, "connect_four: (rows+1)*columns must fit a 128-bit board (and columns <= 32)");
    const bool wide = (rows + 1) * cols > 64;
    d.game_kind = kC4;
    d.num_distinct_actions = cols;
    d.max_game_length = rows * cols;
    d.obs_size = 3 * rows * cols; d.obs_rank = 3; d.obs_shape[0] = 3; d.obs_shape[1] = rows; d.obs_shape[2] = cols;
    d.state_words = wide ? 4 : 2; d.state_word_bytes = 8;
    C4::Params& p = out->c4;
    p.words = wide ? 4 : 2; p.rows = rows; p.cols = cols; p.k = k; p.ego = ego;
    out->c4_std = (rows == 6 && cols == 7 && k == 4);
    out->c4_wide = wide;
    osg_u128 board = 0, top = 0;
    for (int c = 0; c < cols; ++c) {
      board |= ((static_cast<osg_u128>(1) << rows) - 1) << (c * (rows + 1));
      top |= static_cast<osg_u128>(1) << (c * (rows + 1) + rows - 1);
    }
    p.board = static_cast<uint64_t>(board); p.board_hi = static_cast<uint64_t>(board >> 64);
    p.top = static_cast<uint64_t>(top); p.top_hi = static_cast<uint64_t>(top >> 64);
  } else if (name == "hex") {
    int bs = rd.get_int("board_size", 11);
    int cols = rd.get_int("num_cols", bs), rows = rd.get_int("num_rows", bs);
    bool plain = rd.get_bool("plain_obs_tensor", false);
    std::string rep = rd.get_str("string_rep", "standard");
    bool swap = rd.get_bool("swap", false);
    if (!rd.finish()) return set_error(OSG_ERR_INVALID, rd.err);
    if (rep != "standard" && rep != "explicit") return set_error(OSG_ERR_INVALID, "Invalid string_rep " + rep);
    if (cols < 1 || rows < 1) return set_error(OSG_ERR_INVALID, "hex: bad dimensions");
    int cells = cols * rows;
    // the planes hold up to 384 cells (19 x 19 = 361, with the swap action 362 ids); a row shift must stay below a
    // word (cols <= 31).  The reference has no upper bound (hex.cc:47-56): larger boards stay unsupported here.
    if (cells + (swap ? 1 : 0) > 32 * 12 || cols > 31)
      return set_error(OSG_ERR_UNSUPPORTED, "hex: boards above 384 cells (or wider than 31 columns) have no device layout");
    if (swap && cols > rows)
      return set_error(OSG_ERR_UNSUPPORTED, "hex: swap on a board with more columns than rows writes past the board in the "
                                            "reference (hex.cc:236-241: mirrored_move = c * num_cols + r can exceed the cells)");
    if (plain && cols != rows)
      return set_error(OSG_ERR_UNSUPPORTED, "hex: plain_obs_tensor on a non-square board indexes out of "
                                            "bounds in the reference (hex.cc:382-387)");
    d.game_kind = kHex;
    d.num_distinct_actions = cells + (swap ? 1 : 0);
    d.max_game_length = cells;
    d.obs_rank = 3; d.obs_shape[0] = plain ? 3 : 9; d.obs_shape[1] = cols; d.obs_shape[2] = rows;
    d.obs_size = d.obs_shape[0] * cells;
    // plane words: enough for every action id (the swap action is id `cells`), rounded up to an instantiated width
    const int need = (d.num_distinct_actions + 31) / 32;
    out->hex_nw = need <= 4 ? need : (need <= 6 ? 6 : (need <= 8 ? 8 : 12));
    if (need <= 4 && (cells + 31) / 32 < need) out->hex_nw = need;  // (cells a multiple of 32 with swap: one more word)
    out->hex_explicit = rep == "explicit";
    // (round 5) planes whose last word has at least five spare bits — hex(9), the default 11 x 11, 13 x 13, 15 x 15,
    // 19 x 19 ... — carry the meta word in those bits: 4 * hex_nw words per state instead of 4 * hex_nw + 1
    // (HexT::folded; OSG_HEX_FOLD=0 keeps the separate meta word)
    const char* fold_env = std::getenv("OSG_HEX_FOLD");
    out->hex_fold = cells <= 32 * out->hex_nw - 5 && !(fold_env && fold_env[0] == '0');
    d.state_words = out->hex_fold ? 4 * out->hex_nw : 4 * out->hex_nw + 1; d.state_word_bytes = 4;
    // only the variant that holds the board: Bits::w has NW words, a larger board would write past it
    out->hex1 = {}; out->hex2 = {}; out->hex3 = {}; out->hex4 = {}; out->hex6 = {}; out->hex8 = {}; out->hex12 = {};
    switch (out->hex_nw) {
      case 1: fill_hex<1>(&out->hex1, cols, rows, swap, plain); break;
      case 2: fill_hex<2>(&out->hex2, cols, rows, swap, plain); break;
      case 3: fill_hex<3>(&out->hex3, cols, rows, swap, plain); break;
      case 4: fill_hex<4>(&out->hex4, cols, rows, swap, plain); break;
      case 6: fill_hex<6>(&out->hex6, cols, rows, swap, plain); break;
      case 8: fill_hex<8>(&out->hex8, cols, rows, swap, plain); break;
      default: fill_hex<12>(&out->hex12, cols, rows, swap, plain); break;
    }
    if (out->hex_fold)
      out->hex1.words = out->hex2.words = out->hex3.words = out->hex4.words = out->hex6.words = out->hex8.words = out->hex12.words =
          4 * out->hex_nw;   // (only the variant in use is ever read)
  } else if (name == "kuhn_poker") {
    int n = rd.get_int("players", 2);
    if (!rd.finish()) return set_error(OSG_ERR_INVALID, rd.err);
    if (n < 2 || n > 10) return set_error(OSG_ERR_INVALID, "kuhn_poker: players must be in [2, 10]");
    d.game_kind = kKuhn;
    d.num_players = n;
    d.num_distinct_actions = 2;
    d.max_chance_outcomes = n + 1;
    d.max_game_length = 2 * n - 1;
    d.max_chance_nodes = n;
    d.obs_size = 3 * n + 1; d.obs_rank = 1; d.obs_shape[0] = d.obs_size;     // kuhn_poker.cc:405-410
    d.info_size = 6 * n - 1; d.info_rank = 1; d.info_shape[0] = d.info_size; // kuhn_poker.cc:395-403
    d.min_utility = -2; d.max_utility = (n - 1) * 2;
    d.state_words = 1; d.state_word_bytes = 8;
    out->kuhn.words = 1; out->kuhn.players = n;
  } else if (name == "leduc_poker") {
    int n = rd.get_int("players", 2);
    bool mapping = rd.get_bool("action_mapping", false);
    bool iso = rd.get_bool("suit_isomorphism", false);
    int starter = rd.get_int("starting_player", 0);
    if (!rd.finish()) return set_error(OSG_ERR_INVALID, rd.err);
    if (n < 2 || n > 10) return set_error(OSG_ERR_INVALID, "leduc_poker: players must be in [2, 10]");
    if (starter < 0 || starter >= n) return set_error(OSG_ERR_INVALID, "leduc_poker: bad starting_player");
    int cards = (n + 1) * 2;
    int K = iso ? cards / 2 : cards;
    d.game_kind = kLeduc;
    d.num_players = n;
    d.num_distinct_actions = 3;
    d.max_chance_outcomes = K;
    d.max_game_length = 2 * (3 * n - 2);
    d.max_chance_nodes = 3;
    d.obs_size = n + 2 * K + n; d.obs_rank = 1; d.obs_shape[0] = d.obs_size;                        // leduc_poker.cc:822-831
    d.info_size = n + 2 * K + d.max_game_length * 2; d.info_rank = 1; d.info_shape[0] = d.info_size; // :811-820
    d.min_utility = -13; d.max_utility = (n - 1) * 13;                                               // :833-861
    out->leduc_big = n > 3;   // 4 to 10 players: the five-plane record (osg_game_poker.h LeducT<10>)
    d.state_words = out->leduc_big ? 5 : 2; d.state_word_bytes = 8;
    Leduc::Params& p = out->leduc;
    p.words = d.state_words; p.players = n; p.cards = cards; p.mapping = mapping; p.iso = iso; p.starter = starter;
  } else {
    return set_error(OSG_ERR_UNSUPPORTED, "Unknown game '" + name + "' (device games: tic_tac_toe, "
                     "connect_four, hex, kuhn_poker, leduc_poker)");
  }
  int widest = d.num_distinct_actions > d.max_chance_outcomes ? d.num_distinct_actions : d.max_chance_outcomes;
  d.mask_words = (widest + 31) / 32;
  d.compact_mask_bytes = widest <= 8 ? 1 : (widest <= 16 ? 2 : 4 * d.mask_words);
  // Game::ToString(): name(k=v,...) with the parameters as given, sorted by key.
  std::string canon = name + "(";
  bool first = true;
  for (const auto& kv : rd.given) {
    if (!first) canon += ",";
    canon += kv.first + "=" + kv.second.str();
    first = false;
  }
  canon += ")";
  strncpy(d.canonical, canon.c_str(), sizeof(d.canonical) - 1);
  return OSG_OK;
}

}  // namespace osg

namespace osg { const std::string& last_error_string(); }

extern "C" const char* osg_last_error(void) { return osg::last_error_string().c_str(); }

extern "C" int osg_game_describe(const char* game_string, osg_game_desc* out) {
  if (!out) return osg::set_error(OSG_ERR_INVALID, "null desc");
  osg::GameSpec spec;
  int rc = osg::parse_game(game_string, &spec);
  if (rc != OSG_OK) return rc;
  *out = spec.desc;
  return OSG_OK;
}
