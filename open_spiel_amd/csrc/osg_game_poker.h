// Device-side rules of the two imperfect-information poker games.
//
//   Kuhn  : open_spiel/games/kuhn_poker/kuhn_poker.{h,cc}
//   Leduc : open_spiel/games/leduc_poker/leduc_poker.{h,cc}
#ifndef OSG_GAME_POKER_H_
#define OSG_GAME_POKER_H_

#include <type_traits>

#include "osg_common.h"
#include "osg_game_boards.h"  // GenericObsCursor

namespace osg {

// ===========================================================================
// kuhn_poker (P players, 2 <= P <= 10).  The reference notes that "the move
// history and number of players are sufficient information to specify the
// state" (kuhn_poker.h:83-85) — so the HBM record IS the history, ONE u64:
//   bits 0-4    history length (deals + bets)
//   bits 5-44   card dealt to player p at 5 + 4p (4 bits each)
//   bits 45-63  bet flag of betting action j (0 = pass, 1 = bet), j < 2P-1
// Everything else (first bettor, pot, winner) is recomputed in registers.
// ===========================================================================
struct Kuhn {
  using word_t = uint64_t;
  static constexpr int kMaskW = kMaskWords;
  struct Params {
    int words;  // = 1
    int players;
  };
  struct State {
    uint64_t h;
  };
  OSG_HD static State initial(const Params&) { return {0ull}; }
  OSG_HD static State load(const Params&, const word_t* base, int64_t, int64_t i) { return {base[i]}; }
  OSG_HD static void store(const Params&, word_t* base, int64_t, int64_t i, const State& s) { base[i] = s.h; }
  OSG_HD static int len(const State& s) { return static_cast<int>(s.h & 31ull); }
  OSG_HD static int card(const State& s, int p) { return static_cast<int>((s.h >> (5 + 4 * p)) & 15ull); }
  OSG_HD static uint32_t bets(const State& s) { return static_cast<uint32_t>(s.h >> 45); }
  OSG_HD static int nact(const Params& p, const State& s) { int l = len(s); return l > p.players ? l - p.players : 0; }
  // first_bettor_ (kuhn_poker.cc:195-199): the player of the first bet, or -1.
  OSG_HD static int first_bettor(const State& s) {
    uint32_t b = bets(s);
    return b ? __builtin_ctz(b) : -1;  // the first bet happens within the first P actions
  }
  OSG_HD static bool terminal(const Params& p, const State& s) {  // kuhn_poker.cc:206-227
    int n = nact(p, s), fb = first_bettor(s);
    return fb < 0 ? (n == p.players) : (n == p.players + fb);
  }
  OSG_HD static int current_player(const Params& p, const State& s) {  // kuhn_poker.cc:181-188
    if (terminal(p, s)) return kTerminalPlayer;
    int l = len(s);
    return l < p.players ? kChancePlayer : l % p.players;
  }
  OSG_HD static uint32_t dealt_mask(const Params& p, const State& s) {
    uint32_t m = 0;
    int l = len(s);
    for (int q = 0; q < p.players; ++q)
      if (q < l) m |= 1u << card(s, q);
    return m;
  }
  OSG_HD static Mask legal(const Params& p, const State& s) {  // kuhn_poker.cc:231-242
    Mask m;
    if (terminal(p, s)) return m;
    if (len(s) < p.players) m.w[0] = ~dealt_mask(p, s) & ((1u << (p.players + 1)) - 1u);
    else m.w[0] = 3u;
    return m;
  }
  OSG_HD static double chance_prob(const Params& p, const State& s, int) {  // kuhn_poker.cc:329-337
    return 1.0 / (p.players + 1 - len(s));
  }
  OSG_HD static void apply(const Params& p, State& s, int a) {  // kuhn_poker.cc:190-229
    int l = len(s);
    if (l < p.players) s.h |= static_cast<uint64_t>(a) << (5 + 4 * l);
    else s.h |= static_cast<uint64_t>(a & 1) << (45 + (l - p.players));
    s.h = (s.h & ~31ull) | static_cast<uint64_t>(l + 1);
  }
  OSG_HD static bool did_bet(const Params& p, const State& s, int q) {  // DidBet, kuhn_poker.cc:339-349
    int fb = first_bettor(s);
    uint32_t b = bets(s);
    if (fb < 0) return false;
    if (q == fb) return true;
    if (q > fb) return (b >> q) & 1u;
    return (b >> (p.players + q)) & 1u;
  }
  OSG_HD static int winner(const Params& p, const State& s) {
    int fb = first_bettor(s);
    int best_card = -1, best = -1;
    for (int q = 0; q < p.players; ++q) {
      if (fb >= 0 && !did_bet(p, s, q)) continue;  // only bettors contest a bet pot
      int c = card(s, q);
      if (c > best_card) { best_card = c; best = q; }
    }
    return best;
  }
  OSG_HD static int outcome_code(const Params&, const State&) { return 7; }
  OSG_HD static void returns(const Params& p, const State& s, double* out) {  // kuhn_poker.cc:272-283
    if (!terminal(p, s)) {
      for (int q = 0; q < p.players; ++q) out[q] = 0.0;
      return;
    }
    int pot = p.players + __builtin_popcount(bets(s));
    int w = winner(p, s);
    for (int q = 0; q < p.players; ++q) {
      int paid = did_bet(p, s, q) ? 2 : 1;
      out[q] = (q == w) ? static_cast<double>(pot - paid) : static_cast<double>(-paid);
    }
  }
  // ante_[q] (kuhn_poker.cc:196-199): 1 + one chip per bet made so far by q.
  OSG_HD static int contribution(const Params& p, const State& s, int q) {
    uint32_t b = bets(s);
    int c = 1;
    for (int j = q; j < 2 * p.players - 1; j += p.players) c += (b >> j) & 1u;
    return c;
  }
  // KuhnObserver::WriteTensor, kuhn_poker.cc:72-107.
  //   which 1 (information state): player[P] | private_card[P+1] | betting[2P-1, 2]
  //   which 0 (observation)      : player[P] | private_card[P+1] | pot_contribution[P]
  OSG_HD static float obs_at(const Params& p, const State& s, int player, int which, int idx) {
    const int P = p.players;
    int l = len(s);
    if (idx < P) return idx == player ? 1.0f : 0.0f;
    idx -= P;
    if (idx < P + 1) return (l > player && card(s, player) == idx) ? 1.0f : 0.0f;
    idx -= P + 1;
    if (which == 1) {
      int j = idx >> 1, bit = idx & 1;
      if (j >= nact(p, s)) return 0.0f;
      return static_cast<int>((bets(s) >> j) & 1u) == bit ? 1.0f : 0.0f;
    }
    return static_cast<float>(contribution(p, s, idx));
  }
  // The tensor walker: every entry but the pot contributions is 0 or 1, so the row is built ONCE as a bit set
  // (player, private card, the betting pairs) and an entry is a bit test (obs_at re-derives its piece per float).
  struct ObsCursor {
    uint64_t bits;  // 6 P - 1 <= 59 entries for P <= 10
    int idx, nbits;
    OSG_HD void init(const Params& p, const State& s, int player, int which, int idx0) {
      const int P = p.players;
      idx = idx0;
      bits = 1ull << player;
      if (len(s) > player) bits |= 1ull << (P + card(s, player));
      nbits = 2 * P + 1;
      if (which == 1) {  // betting[2P-1, 2]: action j is "pass" (entry 2j) or "bet" (entry 2j + 1)
        const uint32_t b = bets(s);
        const int n = nact(p, s);
        for (int j = 0; j < n; ++j) bits |= 1ull << (nbits + 2 * j + static_cast<int>((b >> j) & 1u));
        nbits += 2 * (2 * P - 1);
      }
    }
    OSG_HD float next(const Params& p, const State& s, int, int) {
      const int k = idx++;
      if (k < nbits) return static_cast<float>((bits >> k) & 1ull);
      return static_cast<float>(contribution(p, s, k - nbits));  // observation tensor: pot_contribution[P]
    }
  };
};

// ===========================================================================
// leduc_poker (2 or 3 players on the device; action_mapping / suit_isomorphism /
// starting_player supported).  The reference keeps ~20 small integers per state
// (leduc_poker.h:173-211); they are packed into TWO u64 planes:
//   word0: cur_player+1 (3b) | calls (2) | raises (2) | round-1 (1) | stakes (4) |
//          pot (6) | public+1 (4) | deck mask (8) | dealt (2) | remaining (2) |
//          folded mask (3) | winner mask (3) | num_winners (2) | ante[3] (4 each) |
//          private+1 [3] (4 each)                                      = 66 -> see below
//   word1: seq1 len (3) + 7 x 2b | seq2 len (3) + 7 x 2b               = 34 bits
// (private cards live in word1's upper bits to keep word0 at 54 bits.)
// money_[p] is always 100 - ante_[p] (+ pot share at the end), so it is not stored.
// ===========================================================================
// kP = the players the record holds: 3 — the two-plane layout above (2 and 3 players: every kernel, the solvers) — or
// 10 — five planes (4 to 10 players, leduc_poker.cc:49-50; the batch entry points):
//   word0: cur_player+1 (4b) | calls (4) | raises (2) | round-1 (1) | stakes (4) | pot (8) | public+1 (5) | dealt (4) |
//          remaining (4) | num_winners (4) | deck mask (22)                                            = 62 bits
//   word1: ante[10] (4 each) | folded mask (10) | winner mask (10)                                    = 60
//   word2: private+1 [10] (5 each) | seq1 len (5) | seq2 len (5)                                      = 60
//   word3: round 1's moves, 2 bits each (<= 3 P - 2 = 28)        word4: round 2's
struct LeducParams {
  int words;  // = 2 (kP = 3) or 5 (kP = 10)
  int players, cards, mapping, iso, starter;
};
template <int kP>
struct LeducT {
  using word_t = uint64_t;
  static constexpr int kMaskW = kMaskWords;
  static constexpr bool kBig = kP > 3;
  using pk_t = typename std::conditional<kBig, uint64_t, uint32_t>::type;   // the per-player / per-round packs
  static constexpr int kPrivBits = kBig ? 5 : 4;
  using Params = LeducParams;
  struct State {
    int cur;          // -1 chance, else player
    int calls, raises, round, stakes, pot, pub, dealt, remaining, nwin;
    uint32_t deck;    // bit c set = card c still in the deck
    uint32_t folded, winners;
    // Small per-player / per-round vectors are kept bit-packed in scalars and read through the
    // accessors below: a runtime-indexed C array would be demoted from registers to scratch memory.
    pk_t ante_pk;   // ante of player q: 4 bits at 4q
    pk_t priv_pk;   // private card of player q, plus one (0 = none): kPrivBits bits at kPrivBits * q
    pk_t seq0, seq1;  // moves of round 1 / round 2, 2 bits each
    int len0, len1;
  };
  OSG_HD static int ante(const State& s, int q) { return static_cast<int>((s.ante_pk >> (4 * q)) & 15u); }
  OSG_HD static int priv(const State& s, int q) {
    return static_cast<int>((s.priv_pk >> (kPrivBits * q)) & ((1u << kPrivBits) - 1u)) - 1;
  }
  OSG_HD static pk_t seq(const State& s, int r) { return r == 0 ? s.seq0 : s.seq1; }
  OSG_HD static int seqlen(const State& s, int r) { return r == 0 ? s.len0 : s.len1; }
  static constexpr int kNone = -1;

  OSG_HD static State initial(const Params& p) {  // leduc_poker.cc:241-286
    State s;
    s.cur = kChancePlayer; s.calls = 0; s.raises = 0; s.round = 1; s.stakes = 1;
    s.pot = p.players; s.pub = kNone; s.dealt = 0; s.remaining = p.players; s.nwin = 0;
    s.deck = (1u << p.cards) - 1u; s.folded = 0; s.winners = 0;
    s.ante_pk = static_cast<pk_t>(kBig ? 0x1111111111ull : 0x111ull);  // everyone antes 1
    s.priv_pk = 0u;
    s.seq0 = s.seq1 = 0; s.len0 = s.len1 = 0;
    return s;
  }
  OSG_HD static State unpack5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e) {   // the five-plane record
    State s;
    s.cur = static_cast<int>(a & 15ull) - 1;            a >>= 4;
    s.calls = static_cast<int>(a & 15ull);              a >>= 4;
    s.raises = static_cast<int>(a & 3ull);              a >>= 2;
    s.round = static_cast<int>(a & 1ull) + 1;           a >>= 1;
    s.stakes = static_cast<int>(a & 15ull);             a >>= 4;
    s.pot = static_cast<int>(a & 255ull);               a >>= 8;
    s.pub = static_cast<int>(a & 31ull) - 1;            a >>= 5;
    s.dealt = static_cast<int>(a & 15ull);              a >>= 4;
    s.remaining = static_cast<int>(a & 15ull);          a >>= 4;
    s.nwin = static_cast<int>(a & 15ull);               a >>= 4;
    s.deck = static_cast<uint32_t>(a & 0x3FFFFFull);
    s.ante_pk = static_cast<pk_t>(b & 0xFFFFFFFFFFull);
    s.folded = static_cast<uint32_t>((b >> 40) & 0x3FFull);
    s.winners = static_cast<uint32_t>((b >> 50) & 0x3FFull);
    s.priv_pk = static_cast<pk_t>(c & 0x3FFFFFFFFFFFFull);
    s.len0 = static_cast<int>((c >> 50) & 31ull);
    s.len1 = static_cast<int>((c >> 55) & 31ull);
    s.seq0 = static_cast<pk_t>(d);
    s.seq1 = static_cast<pk_t>(e);
    return s;
  }
  OSG_HD static void pack5(const State& s, uint64_t* w) {
    w[0] = static_cast<uint64_t>(s.cur + 1) | (static_cast<uint64_t>(s.calls) << 4) | (static_cast<uint64_t>(s.raises) << 8) |
           (static_cast<uint64_t>(s.round - 1) << 10) | (static_cast<uint64_t>(s.stakes) << 11) |
           (static_cast<uint64_t>(s.pot) << 15) | (static_cast<uint64_t>(s.pub + 1) << 23) |
           (static_cast<uint64_t>(s.dealt) << 28) | (static_cast<uint64_t>(s.remaining) << 32) |
           (static_cast<uint64_t>(s.nwin) << 36) | (static_cast<uint64_t>(s.deck) << 40);
    w[1] = static_cast<uint64_t>(s.ante_pk) | (static_cast<uint64_t>(s.folded) << 40) | (static_cast<uint64_t>(s.winners) << 50);
    w[2] = static_cast<uint64_t>(s.priv_pk) | (static_cast<uint64_t>(s.len0) << 50) | (static_cast<uint64_t>(s.len1) << 55);
    w[3] = static_cast<uint64_t>(s.seq0);
    w[4] = static_cast<uint64_t>(s.seq1);
  }
  OSG_HD static State unpack(uint64_t a, uint64_t b) {
    State s;
    s.cur = static_cast<int>(a & 7ull) - 1;            a >>= 3;
    s.calls = static_cast<int>(a & 3ull);               a >>= 2;
    s.raises = static_cast<int>(a & 3ull);              a >>= 2;
    s.round = static_cast<int>(a & 1ull) + 1;           a >>= 1;
    s.stakes = static_cast<int>(a & 15ull);             a >>= 4;
    s.pot = static_cast<int>(a & 63ull);                a >>= 6;
    s.pub = static_cast<int>(a & 15ull) - 1;            a >>= 4;
    s.deck = static_cast<uint32_t>(a & 255ull);         a >>= 8;
    s.dealt = static_cast<int>(a & 3ull);               a >>= 2;
    s.remaining = static_cast<int>(a & 3ull);           a >>= 2;
    s.folded = static_cast<uint32_t>(a & 7ull);         a >>= 3;
    s.winners = static_cast<uint32_t>(a & 7ull);        a >>= 3;
    s.nwin = static_cast<int>(a & 3ull);                a >>= 2;
    s.ante_pk = static_cast<pk_t>(a & 0xFFFull);
    s.len0 = static_cast<int>(b & 7ull);              b >>= 3;
    s.seq0 = static_cast<pk_t>(b & 0x3FFFull);         b >>= 14;
    s.len1 = static_cast<int>(b & 7ull);              b >>= 3;
    s.seq1 = static_cast<pk_t>(b & 0x3FFFull);         b >>= 14;
    s.priv_pk = static_cast<pk_t>(b & 0xFFFull);
    return s;
  }
  OSG_HD static void pack(const State& s, uint64_t& a, uint64_t& b) {
    a = 0; b = 0;
    int sh = 0;
    auto put = [&](uint64_t& w, uint64_t v, int bits) { w |= v << sh; sh += bits; };
    put(a, static_cast<uint64_t>(s.cur + 1), 3);
    put(a, s.calls, 2); put(a, s.raises, 2); put(a, s.round - 1, 1); put(a, s.stakes, 4);
    put(a, s.pot, 6); put(a, static_cast<uint64_t>(s.pub + 1), 4); put(a, s.deck, 8);
    put(a, s.dealt, 2); put(a, s.remaining, 2); put(a, s.folded, 3); put(a, s.winners, 3);
    put(a, s.nwin, 2);
    put(a, s.ante_pk, 12);
    sh = 0;
    put(b, s.len0, 3); put(b, s.seq0, 14); put(b, s.len1, 3); put(b, s.seq1, 14);
    put(b, s.priv_pk, 12);
  }
  OSG_HD static State load(const Params&, const word_t* base, int64_t n, int64_t i) {
    if constexpr (kBig) return unpack5(base[i], base[n + i], base[2 * n + i], base[3 * n + i], base[4 * n + i]);
    else return unpack(base[i], base[n + i]);
  }
  OSG_HD static void store(const Params&, word_t* base, int64_t n, int64_t i, const State& s) {
    if constexpr (kBig) {
      uint64_t w[5];
      pack5(s, w);
#pragma unroll
      for (int k = 0; k < 5; ++k) base[k * n + i] = w[k];
    } else {
      uint64_t a, b;
      pack(s, a, b);
      base[i] = a;
      base[n + i] = b;
    }
  }
  OSG_HD static bool round_over(const State& s) {  // ReadyForNextRound, leduc_poker.cc:680-683
    return (s.raises == 0 && s.calls == s.remaining) || (s.raises > 0 && s.calls == s.remaining - 1);
  }
  OSG_HD static bool terminal(const Params&, const State& s) {  // leduc_poker.cc:498-500
    return s.remaining == 1 || (s.round == 2 && round_over(s));
  }
  OSG_HD static int current_player(const Params& p, const State& s) {  // leduc_poker.cc:288-294
    return terminal(p, s) ? kTerminalPlayer : s.cur;
  }
  OSG_HD static int deck_size(const State& s) { return __builtin_popcount(s.deck); }
  OSG_HD static Mask legal(const Params& p, const State& s) {  // leduc_poker.cc:416-457
    Mask m;
    if (terminal(p, s)) return m;
    if (s.cur == kChancePlayer) {
      if (p.iso) {
        for (int c = 0; c < p.cards / 2; ++c)
          if ((s.deck >> (2 * c)) & 3u) m.w[0] |= 1u << c;
      } else {
        m.w[0] = s.deck;
      }
      return m;
    }
    if (p.mapping) { m.w[0] = 7u; return m; }
    if (s.stakes > ante(s, s.cur)) m.w[0] |= 1u;  // fold only under pressure
    m.w[0] |= 2u;                                 // call / check
    if (s.raises < 2) m.w[0] |= 4u;               // raise
    return m;
  }
  OSG_HD static double chance_prob(const Params& p, const State& s, int outcome) {  // leduc_poker.cc:546-571
    double pr = 1.0 / deck_size(s);
    if (p.iso && ((s.deck >> (2 * outcome)) & 3u) == 3u) return pr * 2;
    return pr;
  }
  OSG_HD static int take_card(const Params& p, State& s, int move) {
    if (p.iso) {
      if ((s.deck >> (2 * move)) & 1u) s.deck &= ~(1u << (2 * move));
      else s.deck &= ~(1u << (2 * move + 1));
      return move;
    }
    s.deck &= ~(1u << move);
    return move;  // deck_[move] == move while present
  }
  static constexpr int kDevicePlayers = kP;  // the players this record holds (osg_game_spec.hip picks the layout)
  OSG_HD static int next_actor(const Params& p, const State& s) {  // NextPlayer, leduc_poker.cc:573-591
    // (no runtime modulo — an integer division costs ~40 instructions — and a fixed trip count)
    const int P = p.players;
    int from = s.cur;
    if (s.cur == kChancePlayer) from = p.starter == 0 ? P - 1 : p.starter - 1;
    int found = 0;
    bool have = false;
#pragma unroll
    for (int i = 1; i <= kDevicePlayers; ++i) {
      int q = from + i;
      q = q >= P ? q - P : q;
      const bool ok = i <= P && !have && !((s.folded >> q) & 1u);
      found = ok ? q : found;
      have |= ok;
    }
    return found;
  }
  OSG_HD static int hand_rank(const Params& p, const State& s, int q) {  // RankHand, leduc_poker.cc:593-626
    int lo = s.pub, hi = priv(s, q);
    if (lo > hi) { int t = lo; lo = hi; hi = t; }
    if (p.iso) {
      int n = p.cards / 2;
      return lo == hi ? n * n + lo : hi * n + lo;
    }
    int n = p.cards;
    if ((lo & 1) == 0 && hi == lo + 1) return n * n + lo;
    return (hi / 2) * n + (lo / 2);
  }
  OSG_HD static void showdown(const Params& p, State& s) {  // ResolveWinner, leduc_poker.cc:628-678
    const uint32_t alive = ~s.folded & ((1u << p.players) - 1u);
    if (s.remaining == 1) {  // the one player left: the lowest (only) bit of `alive`
      if (alive) { s.nwin = 1; s.winners = alive & (0u - alive); }
      return;
    }
    int best = -1;
    s.nwin = 0; s.winners = 0;
#pragma unroll
    for (int q = 0; q < kDevicePlayers; ++q) {
      const bool in = q < p.players && ((alive >> q) & 1u);
      const int r = in ? hand_rank(p, s, q) : -2;
      const bool better = r > best, same = in && r == best;
      s.winners = better ? (1u << q) : (same ? s.winners | (1u << q) : s.winners);
      s.nwin = better ? 1 : (same ? s.nwin + 1 : s.nwin);
      best = better ? r : best;
    }
  }
  OSG_HD static void pay(State& s, int q, int amount) {  // Ante, leduc_poker.cc:700-704
    s.pot += amount;
    s.ante_pk += static_cast<pk_t>(amount) << (4 * q);  // an ante never exceeds 13
  }
  OSG_HD static void record(State& s, int move) {
    int r = s.round - 1;
    if (r == 0) { s.seq0 |= static_cast<pk_t>(move) << (2 * s.len0); ++s.len0; }
    else { s.seq1 |= static_cast<pk_t>(move) << (2 * s.len1); ++s.len1; }
  }
  OSG_HD static void advance(const Params& p, State& s, bool may_start_round) {
    if (terminal(p, s)) {
      showdown(p, s);
    } else if (may_start_round && round_over(s)) {  // NewRound, leduc_poker.cc:685-691
      s.round = 2; s.raises = 0; s.calls = 0; s.cur = kChancePlayer;
    } else {
      s.cur = next_actor(p, s);
    }
  }
  OSG_HD static void apply(const Params& p, State& s, int a) {  // DoApplyAction, leduc_poker.cc:298-414
    if (s.cur == kChancePlayer) {
      if (s.dealt < p.players) {  // SetPrivate, :706-727
        s.priv_pk |= static_cast<pk_t>(take_card(p, s, a) + 1) << (kPrivBits * s.dealt);
        ++s.dealt;
        if (s.dealt == p.players) s.cur = p.starter;
      } else {
        s.pub = take_card(p, s, a);
        s.cur = next_actor(p, s);
      }
      return;
    }
    if (p.mapping) {  // :333-345
      if (a == 0 && s.stakes <= ante(s, s.cur)) a = 1;
      else if (a == 2 && s.raises >= 2) a = 1;
    }
    // The three actions as selects over the fields they touch, then ONE advance() (fold / call may end the round,
    // a raise never does): inlining advance() — showdown, next actor — once per action tripled the code.
    const bool fold = a == 0, call = a == 1, raise = !fold && !call;
    record(s, fold ? 0 : (call ? 1 : 2));
    s.folded |= fold ? 1u << s.cur : 0u;
    s.remaining -= fold ? 1 : 0;
    const int to_call = s.stakes - ante(s, s.cur);
    const int bump = s.round == 1 ? 2 : 4;  // leduc_poker.h:65-66
    // call: the difference to the stakes; raise: call first (if anything is owed), then the raise amount
    const int paid = call ? to_call : (raise ? (to_call > 0 ? to_call : 0) + bump : 0);
    pay(s, s.cur, paid);
    s.stakes += raise ? bump : 0;
    s.raises += raise ? 1 : 0;
    s.calls = raise ? 0 : s.calls + (call ? 1 : 0);
    advance(p, s, !raise);
  }
  OSG_HD static int outcome_code(const Params&, const State&) { return 7; }
  // Returns = money - 100 (leduc_poker.cc:502-514) with money = 100 - ante, plus
  // pot / num_winners for winners (added in double, :673) — same operation order.
  OSG_HD static void returns(const Params& p, const State& s, double* out) {
    bool term = terminal(p, s);
    // After ResolveWinner the reference zeroes pot_; we keep pot and recompute the share.
    for (int q = 0; q < p.players; ++q) {
      if (!term) { out[q] = 0.0; continue; }
      double money = static_cast<double>(100 - ante(s, q));
      if ((s.winners >> q) & 1u) money += static_cast<double>(s.pot) / s.nwin;
      out[q] = money - 100.0;
    }
  }
  // LeducObserver::WriteTensor, leduc_poker.cc:103-192:
  //   player[P] | private_card[K] | community_card[K] | betting[2, 3P-2, 2] (info)
  //                                                   | pot_contribution[P] (obs)
  OSG_HD static float obs_at(const Params& p, const State& s, int player, int which, int idx) {
    const int P = p.players, K = p.iso ? p.cards / 2 : p.cards;
    if (idx < P) return idx == player ? 1.0f : 0.0f;
    idx -= P;
    if (idx < K) return priv(s, player) == idx ? 1.0f : 0.0f;
    idx -= K;
    if (idx < K) return s.pub == idx ? 1.0f : 0.0f;
    idx -= K;
    if (which == 1) {
      const int bets = 3 * P - 2;
      int r = idx / (bets * 2), rem = idx - r * bets * 2;
      int i = rem >> 1, bit = rem & 1;
      if (i >= seqlen(s, r)) return 0.0f;
      int mv = static_cast<int>((seq(s, r) >> (2 * i)) & 3u);
      return (mv == 1 && bit == 0) || (mv == 2 && bit == 1) ? 1.0f : 0.0f;  // call 10, raise 01
    }
    return static_cast<float>(ante(s, idx));
  }
  // The tensor walker: every entry but the pot contributions is 0 or 1, so the row is built ONCE as a bit set
  // (player, private card, community card, the two betting rounds' call / raise pairs) and an entry is a bit
  // test — ~120 instructions per row instead of ~30 per float through obs_at.
  struct BitCursor {
    uint64_t bits;
    int idx, nbits;
    OSG_HD void init(const Params& p, const State& s, int player, int which, int idx0) {
      const int P = p.players, K = p.iso ? p.cards / 2 : p.cards;
      idx = idx0;
      bits = 1ull << player;
      const int pc = priv(s, player);
      if (pc >= 0) bits |= 1ull << (P + pc);
      if (s.pub >= 0) bits |= 1ull << (P + K + s.pub);
      nbits = P + 2 * K;
      if (which == 1) {
        const int bets = 3 * P - 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const uint32_t q = static_cast<uint32_t>(seq(s, r));
          const int len = seqlen(s, r);
          const int base = P + 2 * K + r * bets * 2;
          for (int i = 0; i < len; ++i) {
            const uint32_t mv = (q >> (2 * i)) & 3u;               // 1 call -> "10", 2 raise -> "01"
            if (mv == 1) bits |= 1ull << (base + 2 * i);
            else if (mv == 2) bits |= 1ull << (base + 2 * i + 1);
          }
        }
        nbits += 2 * bets * 2;
      }
    }
    OSG_HD float next(const Params&, const State& s, int, int) {
      const int k = idx++;
      if (k < nbits) return static_cast<float>((bits >> k) & 1ull);
      return static_cast<float>(ante(s, k - nbits));                // observation tensor: pot_contribution[P]
    }
  };
  // (the information-state row of 4+ players is wider than a 64-bit image: the big record walks obs_at entry by entry)
  using ObsCursor = typename std::conditional<kBig, GenericObsCursor<LeducT<kP>>, BitCursor>::type;
};
using Leduc = LeducT<3>;       // 2 and 3 players
using LeducBig = LeducT<10>;   // 4 to 10 players

}  // namespace osg
#endif  // OSG_GAME_POKER_H_
