// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see spiel_oracle.h).
// extern "C" surface so tests/, bench.py's cpu_baseline leg and
// __graft_entry__.smoke() can drive the oracle through ctypes.
//
// This one driver builds twice:
//   * default: over the restatement (spiel_oracle.h, namespace osg_oracle) -> liboracle.so;
//   * -DOSGO_GENUINE_REFERENCE: over the GENUINE reference classes, compiled unmodified from
//     /root/reference against the private abseil / nlohmann stand-ins in oracle/ref_shim
//     (recipe: oracle/Makefile.ref) -> oracle/_ref/libspiel_ref.so.  Same entry points, same
//     seeded drivers, so every deterministic result of the restatement can be compared with
//     the real implementation call for call (tests/test_oracle_vs_reference.py).  The entry
//     points that need the restatement's replay hooks (counter-stream MCTS, mini-batch
//     MCCFR) report an error in that build: the reference has no such hooks.
#include <map>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <thread>
#include <unordered_map>

#ifdef OSGO_GENUINE_REFERENCE
#include "open_spiel/algorithms/best_response.h"
#include "open_spiel/algorithms/cfr.h"
#include "open_spiel/algorithms/cfr_br.h"
#include "open_spiel/algorithms/expected_returns.h"
#include "open_spiel/algorithms/external_sampling_mccfr.h"
#include "open_spiel/algorithms/mcts.h"
#include "open_spiel/algorithms/outcome_sampling_mccfr.h"
#include "open_spiel/algorithms/tabular_exploitability.h"
#include "open_spiel/games/kuhn_poker/kuhn_poker.h"
#include "open_spiel/observer.h"
#include "open_spiel/policy.h"
#include "open_spiel/spiel.h"
#include "open_spiel/spiel_utils.h"

namespace osgo_adapt {
// The counter-based generator of spiel_oracle_core.cpp (splitmix64 keyed by seed / stream /
// sub-stream), restated here so that this build draws the same playouts as the restatement.
inline uint64_t Mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
struct CounterRng {
  uint64_t s;
  explicit CounterRng(uint64_t seed, uint64_t stream = 0, uint64_t sub = 0) {
    uint64_t a = Mix64(seed + 0x9E3779B97F4A7C15ULL);
    uint64_t b = Mix64(a ^ (stream * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL));
    s = Mix64(b ^ (sub * 0xA0761D6478BD642FULL + 0xE7037ED1A0B428DBULL));
  }
  uint64_t Next() { s += 0x9E3779B97F4A7C15ULL; return Mix64(s); }
  uint32_t Below(uint32_t n) { return static_cast<uint32_t>(((Next() >> 32) * n) >> 32); }
  double Unit() { return static_cast<double>(Next() >> 11) * (1.0 / 9007199254740992.0); }
};
[[noreturn]] inline void Fatal(const std::string& msg) { throw std::runtime_error(msg); }
// SpielFatalError -> exception, as the reference's own Python bindings do (pyspiel.cc:831-837).
inline void ThrowingHandler(const std::string& msg) { throw std::runtime_error(msg); }
struct InstallHandler {
  InstallHandler() { open_spiel::SetErrorHandler(&ThrowingHandler); }
};
static InstallHandler g_install_handler;
}  // namespace osgo_adapt
#define ORACLE_CHECK(cond) \
  do { if (!(cond)) ::osgo_adapt::Fatal(std::string(__FILE__) + ":" + std::to_string(__LINE__) + " CHECK failed: " #cond); } while (0)

using namespace open_spiel;
using namespace open_spiel::algorithms;
using osgo_adapt::CounterRng;
using osgo_adapt::Fatal;

namespace {
void WriteTensor(const State& s, int which, int p, float* out, int n) {
  if (which == 0) s.ObservationTensor(p, absl::MakeSpan(out, n));
  else s.InformationStateTensor(p, absl::MakeSpan(out, n));
}
std::string ParamsString(const Game& game) { return GameParametersToString(game.GetParameters()); }
bool HasChance(const Game& game) { return game.GetType().chance_mode != GameType::ChanceMode::kDeterministic; }
// The reference raises for tensors a game does not provide (spiel.h:1085-1090); the restatement reports size 0.
int InfoSize(const Game& game) { return game.GetType().provides_information_state_tensor ? game.InformationStateTensorSize() : 0; }
std::vector<int> InfoShape(const Game& game) { return game.GetType().provides_information_state_tensor ? game.InformationStateTensorShape() : std::vector<int>{}; }
// cfr.cc:104-125 (CFRAveragePolicy::GetStatePolicyFromInformationStateValues is private).
ActionsAndProbs AverageFromValues(const CFRInfoStateValues& v) {
  std::unordered_map<std::string, CFRInfoStateValues> one{{"k", v}};
  CFRAveragePolicy pol(one, nullptr);
  return pol.GetStatePolicy(std::string("k"));
}
TabularPolicy KuhnOptimal(double alpha) { return kuhn_poker::GetOptimalPolicy(alpha); }
void SetRow(TabularPolicy* pol, const std::string& key, const ActionsAndProbs& ap) { pol->SetStatePolicy(key, ap); }
std::vector<double> ExpReturns(const State& s, const Policy& pol) { return ExpectedReturns(s, pol, -1, /*use_infostate_get_policy=*/false); }
}  // namespace
#else
#include "spiel_oracle.h"

using namespace osg_oracle;

namespace {
void WriteTensor(const State& s, int which, int p, float* out, int n) {
  if (which == 0) s.ObservationTensor(p, out, n);
  else s.InformationStateTensor(p, out, n);
}
std::string ParamsString(const Game& game) { return game.ParametersString(); }
bool HasChance(const Game& game) { return game.HasChance(); }
int InfoSize(const Game& game) { return game.InformationStateTensorSize(); }
std::vector<int> InfoShape(const Game& game) { return game.InformationStateTensorShape(); }
ActionsAndProbs AverageFromValues(const CFRInfoStateValues& v) { return CFRAveragePolicy::FromValues(v); }
TabularPolicy KuhnOptimal(double alpha) { return KuhnOptimalPolicy(alpha); }
void SetRow(TabularPolicy* pol, const std::string& key, const ActionsAndProbs& ap) { pol->Table()[key] = ap; }
std::vector<double> ExpReturns(const State& s, const Policy& pol) { return ExpectedReturns(s, pol); }
}  // namespace
#endif

namespace {
thread_local std::string g_err;
struct GameH {
  std::shared_ptr<const Game> game;
};
struct StateH {
  std::unique_ptr<State> state;
};
struct CfrH {
  std::shared_ptr<const Game> game;
  std::unique_ptr<CFRSolverBase> cfr;
  std::unique_ptr<ExternalSamplingMCCFRSolver> mccfr;
  std::unique_ptr<OutcomeSamplingMCCFRSolver> osmccfr;
  CFRInfoStateValuesTable& Table() {
    return cfr ? cfr->InfoStateValuesTable()
               : (mccfr ? mccfr->InfoStateValuesTable() : osmccfr->InfoStateValuesTable());
  }
  std::shared_ptr<Policy> Average() const {
    return cfr ? cfr->AveragePolicy() : (mccfr ? mccfr->AveragePolicy() : osmccfr->AveragePolicy());
  }
};
int CopyStr(const std::string& s, char* buf, int cap) {
  int n = static_cast<int>(s.size());
  if (buf && cap > 0) {
    int m = std::min(n, cap - 1);
    memcpy(buf, s.data(), m);
    buf[m] = 0;
  }
  return n;
}
template <typename F>
int Guard(F&& f) {
  try {
    return f();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
}  // namespace

extern "C" {

const char* osgo_last_error() { return g_err.c_str(); }

void* osgo_load_game(const char* game_string) {
  try {
    return new GameH{LoadGame(game_string)};
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void osgo_free_game(void* g) { delete static_cast<GameH*>(g); }

// out[0..9]: num_actions, max_chance, players, obs_size, info_size, max_len,
// max_chance_nodes, min_util, max_util, has_chance
int osgo_game_info(void* g, double* out) {
  return Guard([&] {
    const Game& game = *static_cast<GameH*>(g)->game;
    out[0] = game.NumDistinctActions();
    out[1] = game.MaxChanceOutcomes();
    out[2] = game.NumPlayers();
    out[3] = game.ObservationTensorSize();
    out[4] = InfoSize(game);
    out[5] = game.MaxGameLength();
    out[6] = game.MaxChanceNodesInHistory();
    out[7] = game.MinUtility();
    out[8] = game.MaxUtility();
    out[9] = HasChance(game) ? 1 : 0;
    return 0;
  });
}
int osgo_game_string(void* g, int which, char* buf, int cap) {
  const Game& game = *static_cast<GameH*>(g)->game;
  return CopyStr(which == 0 ? game.ToString() : ParamsString(game), buf, cap);
}
int osgo_game_shape(void* g, int which, int* out, int cap) {
  const Game& game = *static_cast<GameH*>(g)->game;
  std::vector<int> s = which == 0 ? game.ObservationTensorShape()
                                  : InfoShape(game);
  for (int i = 0; i < static_cast<int>(s.size()) && i < cap; ++i) out[i] = s[i];
  return static_cast<int>(s.size());
}

void* osgo_new_state(void* g) {
  return new StateH{static_cast<GameH*>(g)->game->NewInitialState()};
}
void* osgo_clone_state(void* s) {
  return new StateH{static_cast<StateH*>(s)->state->Clone()};
}
void osgo_free_state(void* s) { delete static_cast<StateH*>(s); }

int osgo_apply(void* s, int64_t a) {
  return Guard([&] {
    static_cast<StateH*>(s)->state->ApplyAction(a);
    return 0;
  });
}
int osgo_current_player(void* s) {
  return static_cast<StateH*>(s)->state->CurrentPlayer();
}
int osgo_is_terminal(void* s) {
  return static_cast<StateH*>(s)->state->IsTerminal() ? 1 : 0;
}
int osgo_legal_actions(void* s, int64_t* out, int cap) {
  return Guard([&] {
    std::vector<Action> la = static_cast<StateH*>(s)->state->LegalActions();
    for (int i = 0; i < static_cast<int>(la.size()) && i < cap; ++i) out[i] = la[i];
    return static_cast<int>(la.size());
  });
}
int osgo_legal_actions_for(void* s, int player, int64_t* out, int cap) {
  return Guard([&] {
    std::vector<Action> la = static_cast<StateH*>(s)->state->LegalActions(player);
    for (int i = 0; i < static_cast<int>(la.size()) && i < cap; ++i) out[i] = la[i];
    return static_cast<int>(la.size());
  });
}
int osgo_returns(void* s, double* out) {
  return Guard([&] {
    std::vector<double> r = static_cast<StateH*>(s)->state->Returns();
    for (size_t i = 0; i < r.size(); ++i) out[i] = r[i];
    return static_cast<int>(r.size());
  });
}
int osgo_chance_outcomes(void* s, int64_t* acts, double* probs, int cap) {
  return Guard([&] {
    ActionsAndProbs ap = static_cast<StateH*>(s)->state->ChanceOutcomes();
    for (int i = 0; i < static_cast<int>(ap.size()) && i < cap; ++i) {
      acts[i] = ap[i].first;
      probs[i] = ap[i].second;
    }
    return static_cast<int>(ap.size());
  });
}
int osgo_tensor(void* s, int which, int player, float* out, int cap) {
  return Guard([&] {
    const State& st = *static_cast<StateH*>(s)->state;
    std::vector<float> t =
        which == 0 ? st.ObservationTensor(player) : st.InformationStateTensor(player);
    if (static_cast<int>(t.size()) > cap) Fatal("tensor buffer too small");
    std::copy(t.begin(), t.end(), out);
    return static_cast<int>(t.size());
  });
}
// which: 0 ToString, 1 InformationStateString(player), 2 ObservationString,
// 3 HistoryString, 4 ActionToString(player, action)
int osgo_string(void* s, int which, int player, int64_t action, char* buf, int cap) {
  try {
    const State& st = *static_cast<StateH*>(s)->state;
    std::string r;
    switch (which) {
      case 0: r = st.ToString(); break;
      case 1: r = st.InformationStateString(player); break;
      case 2: r = st.ObservationString(player); break;
      case 3: r = st.HistoryString(); break;
      case 4: r = st.ActionToString(player, action); break;
      default: Fatal("bad string selector");
    }
    return CopyStr(r, buf, cap);
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
// SerializeGameAndState (spiel.cc:582-603): the reference's text form of a game + state.  Genuine build only
// (the restatement does not restate wire formats; the product's host mirror does).
int osgo_serialize_game_and_state(void* s, char* buf, int cap) {
#ifdef OSGO_GENUINE_REFERENCE
  try {
    const State& st = *static_cast<StateH*>(s)->state;
    return CopyStr(SerializeGameAndState(*st.GetGame(), st), buf, cap);
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
#else
  (void)s; (void)buf; (void)cap;
  g_err = "wire formats are only available from the genuine reference build";
  return -1;
#endif
}
// Game::MakeObserver(IIGObservationType{public_info, perfect_recall, private_info}) on the state (observer.cc:137-190, the
// games' own MakeObserver): the Observation's tensor, its pieces as "name:d0xd1;...", its string, and Compress()
// (observer.cc:246-309).  private_info: 0 kNone, 1 kSinglePlayer, 2 kAllPlayers; public_info < 0 = no type (the
// game's default observer).  Returns the tensor's length, -2 when the game offers no such observer.  Genuine build only.
int osgo_observer(void* s, int public_info, int perfect_recall, int private_info, int player, float* tensor, int tensor_cap,
                  char* spec, int spec_cap, char* str, int str_cap, unsigned char* comp, int comp_cap, int* comp_len) {
#ifdef OSGO_GENUINE_REFERENCE
  try {
    const State& st = *static_cast<StateH*>(s)->state;
    std::shared_ptr<const Game> game = st.GetGame();
    std::optional<open_spiel::IIGObservationType> type;
    if (public_info >= 0)
      type = open_spiel::IIGObservationType{public_info != 0, perfect_recall != 0,
                                            private_info == 0 ? open_spiel::PrivateInfoType::kNone
                                            : private_info == 1 ? open_spiel::PrivateInfoType::kSinglePlayer
                                                                : open_spiel::PrivateInfoType::kAllPlayers};
    std::shared_ptr<open_spiel::Observer> observer = game->MakeObserver(type, {});
    if (!observer) return -2;
    open_spiel::Observation obs(*game, observer);
    std::string text, pieces;
    if (obs.HasString()) text = obs.StringFrom(st, player);
    int n = 0;
    if (obs.HasTensor()) {
      obs.SetFrom(st, player);
      auto span = obs.Tensor();
      n = static_cast<int>(span.size());
      if (n > tensor_cap) Fatal("tensor buffer too small");
      std::copy(span.begin(), span.end(), tensor);
      for (const auto& info : obs.tensors_info()) {
        pieces += std::string(info.name()) + ":";
        bool first = true;
        for (int d : info.shape()) { pieces += (first ? "" : "x") + std::to_string(d); first = false; }
        pieces += ";";
      }
      const std::string c = obs.Compress();
      if (static_cast<int>(c.size()) > comp_cap) Fatal("compress buffer too small");
      std::memcpy(comp, c.data(), c.size());
      *comp_len = static_cast<int>(c.size());
    } else {
      *comp_len = 0;
      pieces = "-";
    }
    if (CopyStr(pieces, spec, spec_cap) < 0 || CopyStr(text, str, str_cap) < 0) Fatal("string buffer too small");
    return n;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
#else
  (void)s; (void)public_info; (void)perfect_recall; (void)private_info; (void)player; (void)tensor; (void)tensor_cap;
  (void)spec; (void)spec_cap; (void)str; (void)str_cap; (void)comp; (void)comp_cap; (void)comp_len;
  g_err = "general observers are only available from the genuine reference build";
  return -1;
#endif
}
int osgo_history(void* s, int64_t* out, int cap) {
  std::vector<Action> h = static_cast<StateH*>(s)->state->History();
  for (int i = 0; i < static_cast<int>(h.size()) && i < cap; ++i) out[i] = h[i];
  return static_cast<int>(h.size());
}

// ---------------------------------------------------------------------------
// Seeded random playouts with a full per-ply record: the differential-test
// workhorse.  Playout i uses CounterRng(seed, i).  At every ply t (including
// the final position) it records
//   mask  [n, L+1, W] u32  bit-packed LegalActions (chance outcomes at chance
//                          nodes), W = ceil(max(A, max_chance) / 32)
//   cur   [n, L+1]    i8   CurrentPlayer()
//   term  [n, L+1]    u8   IsTerminal()
//   rets  [n, L+1, P] f64  Returns()
//   acts  [n, L]      i16  action applied at ply t, -1 past the end
//   obs   [n, L+1, P, obs_size] f32 (optional, may be NULL)
//   info  [n, L+1, P, info_size] f32 (optional, may be NULL)
// stop[i] (optional) = number of plies to play for playout i (else to the end).
// Chance actions are drawn with SampleAction(ChanceOutcomes(), rng.Unit()),
// player actions uniformly with rng.Below(#legal).  Returns plies played max.
// ---------------------------------------------------------------------------
int osgo_random_playouts(void* g, uint64_t seed, int64_t n, int L, int W,
                         const int32_t* stop, int16_t* acts, uint32_t* mask,
                         int8_t* cur, uint8_t* term, double* rets, float* obs,
                         float* info) {
  return Guard([&] {
    const Game& game = *static_cast<GameH*>(g)->game;
    const int P = game.NumPlayers();
    const int osz = game.ObservationTensorSize();
    const int isz = InfoSize(game);
    int longest = 0;
    for (int64_t i = 0; i < n; ++i) {
      CounterRng rng(seed, static_cast<uint64_t>(i));
      std::unique_ptr<State> s = game.NewInitialState();
      int limit = stop ? stop[i] : L;
      for (int t = 0; t <= L; ++t) {
        const int64_t row = i * (L + 1) + t;
        std::vector<Action> legal = s->LegalActions();
        for (int w = 0; w < W; ++w) mask[row * W + w] = 0;
        for (Action a : legal) mask[row * W + a / 32] |= (1u << (a % 32));
        cur[row] = static_cast<int8_t>(s->CurrentPlayer());
        term[row] = s->IsTerminal();
        std::vector<double> r = s->Returns();
        for (int p = 0; p < P; ++p) rets[row * P + p] = r[p];
        if (obs)
          for (int p = 0; p < P; ++p) WriteTensor(*s, 0, p, obs + (row * P + p) * osz, osz);
        if (info && isz > 0)
          for (int p = 0; p < P; ++p) WriteTensor(*s, 1, p, info + (row * P + p) * isz, isz);
        if (t == L) break;
        if (s->IsTerminal() || t >= limit) {
          acts[i * L + t] = -1;
          continue;
        }
        Action a;
        if (s->IsChanceNode()) {
          a = SampleAction(s->ChanceOutcomes(), rng.Unit()).first;
        } else {
          a = legal[rng.Below(static_cast<uint32_t>(legal.size()))];
        }
        acts[i * L + t] = static_cast<int16_t>(a);
        s->ApplyAction(a);
        longest = std::max(longest, t + 1);
      }
    }
    return longest;
  });
}

// ---------------------------------------------------------------------------
// SURVEY.md 8(d) synthetic benchmark inputs: the CPU side of osg_synth_batch (open_spiel_amd/csrc/osg_kernels.hip
// k_synth), restated call for call so that the batch a benchmark times on the device can be regenerated and
// checked state for state:
//   rng = CounterRng(seed, first + i, "SYNTH"); depth = rng.Below(depth_mod);
//   play `depth` moves from NewInitialState() (chance by SampleAction, players uniformly over LegalActions());
//   a trajectory that is terminal before `depth` moves is thrown away and re-drawn from the same stream
//   (<= 2^14 attempts, then the initial state with depth 0); action = one more draw at the accepted state.
// Per state, for the accepted state ("before") and its successor after ApplyAction(action) ("after"):
//   mask [n, W] u32 LegalActions (chance outcomes at chance nodes), cur [n] i8, term [n] u8, rets [n, P] f64,
//   obs [n, obs_size] u8 = ObservationTensor(player 0) cast to bytes (0 / 1 planes on the boards, small integers
//   in the poker games) — optional.  depth [n] i32, action [n] i16.  `threads` workers over contiguous ranges.
// ---------------------------------------------------------------------------
namespace {
constexpr uint64_t kSynthSub = 0x53594E5448ULL;  // "SYNTH": the sub-stream of osg_synth_batch
Action SynthDraw(const State& s, CounterRng& rng) {
  if (s.IsChanceNode()) return SampleAction(s.ChanceOutcomes(), rng.Unit()).first;
  std::vector<Action> la = s.LegalActions();
  return la[rng.Below(static_cast<uint32_t>(la.size()))];
}
// The accepted state of stream `rng` (positioned at its start): see osgo_synth_batch.
std::unique_ptr<State> SynthState(const Game& game, int depth_mod, CounterRng* rng, int* depth_out) {
  int depth = static_cast<int>(rng->Below(static_cast<uint32_t>(depth_mod)));
  std::unique_ptr<State> s;
  for (int attempt = 0;; ++attempt) {
    s = game.NewInitialState();
    if (attempt >= (1 << 14)) { depth = 0; break; }
    for (int t = 0; t < depth && !s->IsTerminal(); ++t) s->ApplyAction(SynthDraw(*s, *rng));
    if (!s->IsTerminal()) break;
  }
  *depth_out = depth;
  return s;
}
}  // namespace
int osgo_synth_batch(void* g, uint64_t seed, int64_t first, int64_t n, int depth_mod, int threads, int W,
                     int32_t* depth_out, int16_t* action_out,
                     uint32_t* mask0, int8_t* cur0, uint8_t* term0, uint8_t* obs0,
                     uint32_t* mask1, int8_t* cur1, uint8_t* term1, double* rets1, uint8_t* obs1) {
  return Guard([&] {
    const std::string game_string = static_cast<GameH*>(g)->game->ToString();
    threads = std::max(1, threads);
    std::vector<std::string> errors(threads);
    auto work = [&](int w) {
      try {
        // one Game per worker: State objects pin their Game through a shared_ptr (spiel.h:906)
        std::shared_ptr<const Game> game = LoadGame(game_string);
        const int P = game->NumPlayers();
        const int osz = game->ObservationTensorSize();
        std::vector<float> tensor(osz);
        auto record = [&](const State& s, int64_t i, uint32_t* mask, int8_t* cur, uint8_t* term, double* rets, uint8_t* obs) {
          if (mask) {
            for (int k = 0; k < W; ++k) mask[i * W + k] = 0;
            for (Action a : s.LegalActions()) mask[i * W + a / 32] |= (1u << (a % 32));
          }
          if (cur) cur[i] = static_cast<int8_t>(s.CurrentPlayer());
          if (term) term[i] = s.IsTerminal();
          if (rets) {
            std::vector<double> r = s.Returns();
            for (int p = 0; p < P; ++p) rets[i * P + p] = r[p];
          }
          if (obs) {
            WriteTensor(s, 0, 0, tensor.data(), osz);
            for (int k = 0; k < osz; ++k) obs[i * osz + k] = static_cast<uint8_t>(tensor[k]);
          }
        };
        const int64_t lo = n * w / threads, hi = n * (w + 1) / threads;
        for (int64_t i = lo; i < hi; ++i) {
          CounterRng rng(seed, static_cast<uint64_t>(first + i), kSynthSub);
          int depth = 0;
          std::unique_ptr<State> s = SynthState(*game, depth_mod, &rng, &depth);
          const Action a = SynthDraw(*s, rng);
          if (depth_out) depth_out[i] = depth;
          if (action_out) action_out[i] = static_cast<int16_t>(a);
          record(*s, i, mask0, cur0, term0, nullptr, obs0);
          if (mask1 || cur1 || term1 || rets1 || obs1) {
            s->ApplyAction(a);
            record(*s, i, mask1, cur1, term1, rets1, obs1);
          }
        }
      } catch (const std::exception& e) {
        errors[w] = e.what();
      }
    };
    std::vector<std::thread> workers;
    for (int w = 1; w < threads; ++w) workers.emplace_back(work, w);
    work(0);
    for (auto& t : workers) t.join();
    for (const std::string& e : errors)
      if (!e.empty()) Fatal(e);
    return 0;
  });
}

// The tensors of the same synthetic states for any player and either tensor kind (osgo_synth_batch records only
// ObservationTensor(0)): state i of the stream before (after = 0) or after (1) its action, `which` = 0 ObservationTensor / 1
// InformationStateTensor (spiel.cc:908-945 through the game's observer), as bytes [n, size] (every entry on this path
// is a small non-negative integer).  Lets the large-batch tensor kernels meet the CPU entry by entry.
int osgo_synth_tensors(void* g, uint64_t seed, int64_t first, int64_t n, int depth_mod, int threads, int which, int player,
                       int after, uint8_t* out) {
  return Guard([&] {
    const std::string game_string = static_cast<GameH*>(g)->game->ToString();
    threads = std::max(1, threads);
    std::vector<std::string> errors(threads);
    auto work = [&](int w) {
      try {
        std::shared_ptr<const Game> game = LoadGame(game_string);
        const int size = which == 0 ? game->ObservationTensorSize() : InfoSize(*game);
        if (size <= 0) Fatal("osgo_synth_tensors: the game has no such tensor");
        std::vector<float> tensor(size);
        const int64_t lo = n * w / threads, hi = n * (w + 1) / threads;
        for (int64_t i = lo; i < hi; ++i) {
          CounterRng rng(seed, static_cast<uint64_t>(first + i), kSynthSub);
          int depth = 0;
          std::unique_ptr<State> s = SynthState(*game, depth_mod, &rng, &depth);
          if (after) s->ApplyAction(SynthDraw(*s, rng));   // the successor (terminal states among them), as osgo_synth_batch's
          WriteTensor(*s, which, player, tensor.data(), size);
          for (int k = 0; k < size; ++k) {
            if (tensor[k] < 0.0f || tensor[k] > 255.0f || tensor[k] != static_cast<float>(static_cast<uint8_t>(tensor[k])))
              Fatal("osgo_synth_tensors: a tensor entry is not a byte-sized integer");
            out[i * size + k] = static_cast<uint8_t>(tensor[k]);
          }
        }
      } catch (const std::exception& e) {
        errors[w] = e.what();
      }
    };
    std::vector<std::thread> workers;
    for (int w = 1; w < threads; ++w) workers.emplace_back(work, w);
    work(0);
    for (auto& t : workers) t.join();
    for (const std::string& e : errors)
      if (!e.empty()) Fatal(e);
    return 0;
  });
}

// Replays the rollout the HIP kernel performs from a given history: rollout r
// of root `root_index` uses CounterRng(seed, root_index, r) and draws exactly
// as osgo_random_playouts does.  Sums Returns() over n_rollouts into out[P]
// (not divided).  history = actions from the initial state.
int osgo_replay_rollouts(void* g, const int16_t* history, int hist_len,
                         uint64_t seed, uint64_t root_index, int n_rollouts,
                         double* out, int64_t* steps_out) {
  return Guard([&] {
    const Game& game = *static_cast<GameH*>(g)->game;
    std::unique_ptr<State> root = game.NewInitialState();
    for (int t = 0; t < hist_len; ++t) root->ApplyAction(history[t]);
    const int P = game.NumPlayers();
    for (int p = 0; p < P; ++p) out[p] = 0;
    int64_t steps = 0;
    for (int r = 0; r < n_rollouts; ++r) {
      CounterRng rng(seed, root_index, static_cast<uint64_t>(r));
      std::unique_ptr<State> s = root->Clone();
      while (!s->IsTerminal()) {
        if (s->IsChanceNode()) {
          s->ApplyAction(SampleAction(s->ChanceOutcomes(), rng.Unit()).first);
        } else {
          std::vector<Action> legal = s->LegalActions();
          s->ApplyAction(legal[rng.Below(static_cast<uint32_t>(legal.size()))]);
        }
        ++steps;
      }
      std::vector<double> ret = s->Returns();
      for (int p = 0; p < P; ++p) out[p] += ret[p];
    }
    if (steps_out) *steps_out = steps;
    return 0;
  });
}

// ---------------------------------------------------------------------------
// MCTS
// ---------------------------------------------------------------------------
// Runs MCTSBot(RandomRolloutEvaluator(n_rollouts, seed), ...) from `state`.
// out_children: per root child {action, explore_count, total_reward, outcome
// for the root player or NaN}; returns the number of children, best action in
// *best_action, root outcome (for root player; NaN if unsolved) in *root_outcome.
// counter_root >= 0: replay mode, all draws from the device's counter streams for
// that root index with seed `counter_seed` (MCTSBot::UseCounterStreams).
int osgo_mcts_search(void* s, double uct_c, int max_simulations, int n_rollouts,
                     int64_t max_memory_mb, int solve, int seed,
                     int64_t* best_action, double* root_outcome,
                     double* out_children, int cap, int* root_visits,
                     int64_t counter_root, uint64_t counter_seed, int counter_layout, int puct) {
  return Guard([&] {
    const State& st = *static_cast<StateH*>(s)->state;
    auto ev = std::make_shared<RandomRolloutEvaluator>(n_rollouts, seed);
    MCTSBot bot(*st.GetGame(), ev, uct_c, max_simulations, max_memory_mb,
                solve != 0, seed, false, puct ? ChildSelectionPolicy::PUCT : ChildSelectionPolicy::UCT);
#ifdef OSGO_GENUINE_REFERENCE
    if (counter_root >= 0) Fatal("counter-stream replay is a hook of the restatement, not of the reference");
    if (max_memory_mb < 0) Fatal("a direct node budget is a hook of the restatement, not of the reference");
    (void)counter_seed; (void)counter_layout;
#else
    if (counter_root >= 0)
      bot.UseCounterStreams(counter_seed, static_cast<uint64_t>(counter_root), n_rollouts,
                            counter_layout);
#endif
    std::unique_ptr<SearchNode> root = bot.MCTSearch(st);
    const double nan = std::numeric_limits<double>::quiet_NaN();
    *best_action = root->children.empty() ? -1 : root->BestChild().action;
    Player rp = st.CurrentPlayer();
    *root_outcome = (root->outcome.empty() || rp < 0) ? nan : root->outcome[rp];
    if (root_visits) *root_visits = root->explore_count;
    int k = 0;
    for (const SearchNode& c : root->children) {
      if (k >= cap) break;
      out_children[4 * k + 0] = static_cast<double>(c.action);
      out_children[4 * k + 1] = c.explore_count;
      out_children[4 * k + 2] = c.total_reward;
      out_children[4 * k + 3] = (c.outcome.empty() || rp < 0) ? nan : c.outcome[rp];
      ++k;
    }
    return static_cast<int>(root->children.size());
  });
}

// A deterministic stand-in for a value / policy network, in integer arithmetic so that a torch restatement
// of it (tests/test_z5_gpu_mcts_evaluator.py) agrees to the last bit: x = ObservationTensor(current player, or
// player 0 at chance / terminal states) — small non-negative integers —
//   Evaluate: S = sum_i x_i * ((7 i + 3) mod 1009), v = ((S mod 2001) - 1000) / 1024, returns {v, -v / (P - 1), ...}
//   Prior (decision nodes): k_a = 1 + ((sum_i x_i * ((31 i + 17 a + 5) mod 13)) mod 7) for the legal a, p_a = k_a / sum k
//   Prior (chance nodes): ChanceOutcomes(), like every evaluator of the reference (mcts.cc:75-77).
class StubNetEvaluator : public Evaluator {
 public:
  std::vector<double> Evaluate(const State& state) override {
    const std::vector<float> x = Features(state);
    int64_t s = 0;
    for (size_t i = 0; i < x.size(); ++i) s += static_cast<int64_t>(x[i]) * static_cast<int64_t>((7 * i + 3) % 1009);
    const double v = static_cast<double>(s % 2001 - 1000) / 1024.0;  // a power of two: exact, also where a GPU library multiplies by the reciprocal
    const int P = state.NumPlayers();
    std::vector<double> out(P, P > 1 ? -v / (P - 1) : v);
    out[0] = v;
    return out;
  }
  ActionsAndProbs Prior(const State& state) override {
    if (state.IsChanceNode()) return state.ChanceOutcomes();
    const std::vector<float> x = Features(state);
    const std::vector<Action> legal = state.LegalActions();
    std::vector<int64_t> k(legal.size());
    int64_t total = 0;
    for (size_t j = 0; j < legal.size(); ++j) {
      int64_t s = 0;
      for (size_t i = 0; i < x.size(); ++i)
        s += static_cast<int64_t>(x[i]) * static_cast<int64_t>((31 * i + 17 * static_cast<size_t>(legal[j]) + 5) % 13);
      k[j] = 1 + s % 7;
      total += k[j];
    }
    ActionsAndProbs out;
    for (size_t j = 0; j < legal.size(); ++j)
      out.emplace_back(legal[j], static_cast<double>(k[j]) / static_cast<double>(total));
    return out;
  }

 private:
  static std::vector<float> Features(const State& state) {
    const Player cur = state.CurrentPlayer();
    return state.ObservationTensor(cur >= 0 ? cur : 0);
  }
};

// osgo_mcts_search with the stub network as the evaluator (replay mode only: tree-policy draws from the
// device's layout-1 counter streams).  max_nodes <= 0: no node budget.  out_children rows:
// {action, explore_count, total_reward, prior}.
int osgo_mcts_search_stub(void* s, double uct_c, int max_simulations, int max_nodes, int solve, int puct,
                          int dont_return_chance_node, int64_t counter_root, uint64_t counter_seed,
                          int64_t* best_action, double* out_children, int cap, int* root_visits, int* nodes) {
#ifdef OSGO_GENUINE_REFERENCE
  (void)s; (void)uct_c; (void)max_simulations; (void)max_nodes; (void)solve; (void)puct; (void)dont_return_chance_node;
  (void)counter_root; (void)counter_seed; (void)best_action; (void)out_children; (void)cap; (void)root_visits; (void)nodes;
  g_err = "counter-stream replay is a hook of the restatement, not of the reference";
  return -1;
#else
  return Guard([&] {
    const State& st = *static_cast<StateH*>(s)->state;
    auto ev = std::make_shared<StubNetEvaluator>();
    MCTSBot bot(*st.GetGame(), ev, uct_c, max_simulations, max_nodes > 0 ? -static_cast<int64_t>(max_nodes) : 4096,
                solve != 0, 0, false, puct ? ChildSelectionPolicy::PUCT : ChildSelectionPolicy::UCT,
                dont_return_chance_node != 0);
    bot.UseCounterStreams(counter_seed, static_cast<uint64_t>(counter_root), 1, 1);
    bot.UseEvaluatorInReplay();
    std::unique_ptr<SearchNode> root = bot.MCTSearch(st);
    *best_action = root->children.empty() ? -1 : root->BestChild().action;
    if (root_visits) *root_visits = root->explore_count;
    if (nodes) *nodes = bot.LastNodeCount();
    int k = 0;
    for (const SearchNode& c : root->children) {
      if (k >= cap) break;
      out_children[4 * k + 0] = static_cast<double>(c.action);
      out_children[4 * k + 1] = c.explore_count;
      out_children[4 * k + 2] = c.total_reward;
      out_children[4 * k + 3] = c.prior;
      ++k;
    }
    return static_cast<int>(root->children.size());
  });
#endif
}

// Config 4 at full size: roots first .. first + n - 1 of the synthetic stream (osgo_synth_batch's states:
// seed_roots, depth_mod), each searched by MCTSBot(RandomRolloutEvaluator(n_rollouts), uct_c, max_simulations)
// in REPLAY mode — every draw from the device's counter streams (counter_seed, global root index, layout) — on
// `threads` workers.  best [n] i32, visits [n, A] i32, reward [n, A] f64 (zero for actions that are no child).
// Restatement only (the reference has no replay hooks).
int osgo_synth_mcts_replay(void* g, uint64_t seed_roots, int64_t first, int64_t n, int depth_mod, double uct_c,
                           int max_simulations, int n_rollouts, uint64_t counter_seed, int counter_layout,
                           int threads, int32_t* best, int32_t* visits, double* reward) {
#ifdef OSGO_GENUINE_REFERENCE
  (void)g; (void)seed_roots; (void)first; (void)n; (void)depth_mod; (void)uct_c; (void)max_simulations; (void)n_rollouts;
  (void)counter_seed; (void)counter_layout; (void)threads; (void)best; (void)visits; (void)reward;
  g_err = "counter-stream replay is a hook of the restatement, not of the reference";
  return -1;
#else
  return Guard([&] {
    const std::string game_string = static_cast<GameH*>(g)->game->ToString();
    threads = std::max(1, threads);
    std::vector<std::string> errors(threads);
    auto work = [&](int w) {
      try {
        std::shared_ptr<const Game> game = LoadGame(game_string);
        const int A = game->NumDistinctActions();
        for (int64_t i = w; i < n; i += threads) {  // interleaved: search lengths vary with the root's depth
          CounterRng rng(seed_roots, static_cast<uint64_t>(first + i), kSynthSub);
          int depth = 0;
          std::unique_ptr<State> s = SynthState(*game, depth_mod, &rng, &depth);
          auto ev = std::make_shared<RandomRolloutEvaluator>(n_rollouts, 0);
          MCTSBot bot(*game, ev, uct_c, max_simulations, 4096, false, 0, false);
          bot.UseCounterStreams(counter_seed, static_cast<uint64_t>(first + i), n_rollouts, counter_layout);
          std::unique_ptr<SearchNode> root = bot.MCTSearch(*s);
          best[i] = root->children.empty() ? -1 : static_cast<int32_t>(root->BestChild().action);
          for (int a = 0; a < A; ++a) { visits[i * A + a] = 0; reward[i * A + a] = 0.0; }
          for (const SearchNode& c : root->children) {
            visits[i * A + c.action] = c.explore_count;
            reward[i * A + c.action] = c.total_reward;
          }
        }
      } catch (const std::exception& e) {
        errors[w] = e.what();
      }
    };
    std::vector<std::thread> workers;
    for (int w = 1; w < threads; ++w) workers.emplace_back(work, w);
    work(0);
    for (auto& t : workers) t.join();
    for (const std::string& e : errors)
      if (!e.empty()) Fatal(e);
    return 0;
  });
#endif
}

// Self-play of two MCTS bots (mcts_test.cc:45-77); returns via out[P].
int osgo_mcts_selfplay(void* g, double uct_c, int max_simulations, int n_rollouts,
                       int seed, double* out) {
  return Guard([&] {
    const Game& game = *static_cast<GameH*>(g)->game;
    auto ev = std::make_shared<RandomRolloutEvaluator>(n_rollouts, seed);
    MCTSBot b0(game, ev, uct_c, max_simulations, 5, true, seed, false);
    MCTSBot b1(game, ev, uct_c, max_simulations, 5, true, seed + 1, false);
    std::unique_ptr<State> s = game.NewInitialState();
    std::mt19937 chance(seed);
    while (!s->IsTerminal()) {
      if (s->IsChanceNode()) {
        double z = (chance() >> 5) * (1.0 / 134217728.0);
        s->ApplyAction(SampleAction(s->ChanceOutcomes(), z).first);
      } else {
        Action a = (s->CurrentPlayer() == 0 ? b0 : b1).Step(*s);
        s->ApplyAction(a);
      }
    }
    std::vector<double> r = s->Returns();
    for (size_t i = 0; i < r.size(); ++i) out[i] = r[i];
    return 0;
  });
}

// ---------------------------------------------------------------------------
// CFR / MCCFR
// ---------------------------------------------------------------------------
// kind: 0 CFRSolver, 1 CFRPlusSolver, 2 ES-MCCFR(simple), 3 ES-MCCFR(full),
// 4 CFRSolverBase(simultaneous updates, no linear avg, no RM+), 5 OS-MCCFR(epsilon 0.6), 6 CFRBRSolver,
// 16..23 CFRSolverBase with any switch combination (16 + alternating + 2 * linear averaging + 4 * RM+)
void* osgo_cfr_create(void* g, int kind, int seed) {
  try {
    auto* h = new CfrH;
    h->game = static_cast<GameH*>(g)->game;
    if (kind == 0) h->cfr = std::make_unique<CFRSolver>(*h->game);
    else if (kind == 1) h->cfr = std::make_unique<CFRPlusSolver>(*h->game);
    else if (kind == 4) h->cfr = std::make_unique<CFRSolverBase>(*h->game, false, false, false);
    else if (kind == 6) h->cfr = std::make_unique<CFRBRSolver>(*h->game);
    else if (kind >= 16 && kind < 24)  // 16 + alternating_updates + 2 * linear_averaging + 4 * regret_matching_plus
      h->cfr = std::make_unique<CFRSolverBase>(*h->game, (kind & 1) != 0, (kind & 2) != 0, (kind & 4) != 0);
    else if (kind == 5) h->osmccfr = std::make_unique<OutcomeSamplingMCCFRSolver>(
             *h->game, OutcomeSamplingMCCFRSolver::kDefaultEpsilon, seed);
    else h->mccfr = std::make_unique<ExternalSamplingMCCFRSolver>(
             *h->game, seed, kind == 3 ? AverageType::kFull : AverageType::kSimple);
    return h;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void osgo_cfr_free(void* h) { delete static_cast<CfrH*>(h); }
int osgo_cfr_iterate(void* h, int iters) {
  return Guard([&] {
    auto* c = static_cast<CfrH*>(h);
    for (int i = 0; i < iters; ++i) {
      if (c->cfr) c->cfr->EvaluateAndUpdatePolicy();
      else if (c->mccfr) c->mccfr->RunIteration();
      else c->osmccfr->RunIteration();
    }
    return 0;
  });
}
// The DEVICE's mini-batch schedule of external-sampling MCCFR, restated on the
// oracle's solver: trajectories [first, first+count) all read the table as it
// is at the start of the call (trajectory g: traverser g mod P, uniforms from CounterRng(seed, g, 0).Unit() in
// visiting order down to the traverser's first node, then from that generator with its counter jumped per subtree —
// by 1 + b1 inside that node's child b1 down to the traverser's next node, by 16 + 8 b1 + b2 inside that node's child
// b2: the device spreads those subtrees over lanes); their regret / average-policy
// increments (external_sampling_mccfr.cc:167-183) are summed and folded in at
// the end.  With count == 1 this is exactly one UpdateRegrets call.
int osgo_mccfr_minibatch(void* h, uint64_t seed, int64_t first, int64_t count) {
#ifdef OSGO_GENUINE_REFERENCE
  (void)h; (void)seed; (void)first; (void)count;
  g_err = "mini-batch MCCFR replay is a hook of the restatement, not of the reference";
  return -1;
#else
  return Guard([&] {
    auto* c = static_cast<CfrH*>(h);
    ORACLE_CHECK(c->mccfr || c->osmccfr);
    CFRInfoStateValuesTable& table = c->Table();
    const int P = c->game->NumPlayers();
    std::map<std::string, CFRInfoStateValues> delta;  // regrets / cum_policy hold the summed increments
    for (int64_t g = first; g < first + count; ++g) {
      const CFRInfoStateValuesTable frozen = table;
      CounterRng rng(seed, static_cast<uint64_t>(g), 0);
      if (c->mccfr) {
        // the device's streams (csrc/osg_cfr.hip es_stream; osg_common.h Rng::jump_to): stream (seed, g, 0) down to the
        // traverser's first node; its counter jumped by 1 + b1 inside that node's child b1 down to the traverser's next
        // node on that path, by 16 + 8 b1 + b2 inside that node's child b2
        const uint64_t s0 = rng.s;
        const std::function<void(int, int, int)> branch = [&rng, s0](int level, int b1, int b2) {
          const uint64_t id = level == 1 ? 1 + static_cast<uint64_t>(b1) : 16 + 8 * static_cast<uint64_t>(b1) + static_cast<uint64_t>(b2);
          rng.s = s0 + id * 0xD6E8FEB86659FD93ULL;
        };
        c->mccfr->UpdateRegretsWith(*c->game->NewInitialState(), static_cast<Player>(g % P),
                                    [&rng]() { return rng.Unit(); }, &branch);
      } else {
        std::unique_ptr<State> episode = c->game->NewInitialState();
        c->osmccfr->SampleEpisodeWith(episode.get(), static_cast<Player>(g % P), [&rng]() { return rng.Unit(); },
                                      1.0, 1.0, 1.0);
      }
      for (const auto& kv : table) {
        auto fz = frozen.find(kv.first);
        const size_t n = kv.second.legal_actions.size();
        auto it = delta.find(kv.first);
        if (it == delta.end()) {
          CFRInfoStateValues zero(kv.second.legal_actions, 0.0);
          it = delta.emplace(kv.first, zero).first;
        }
        for (size_t a = 0; a < n; ++a) {
          const double r0 = fz == frozen.end() ? ExternalSamplingMCCFRSolver::kInitialTableValues
                                               : fz->second.cumulative_regrets[a];
          const double p0 = fz == frozen.end() ? ExternalSamplingMCCFRSolver::kInitialTableValues
                                               : fz->second.cumulative_policy[a];
          it->second.cumulative_regrets[a] += kv.second.cumulative_regrets[a] - r0;
          it->second.cumulative_policy[a] += kv.second.cumulative_policy[a] - p0;
        }
      }
      // back to the frozen values; rows first seen by this trajectory stay, at their initial values
      for (auto& kv : table) {
        auto fz = frozen.find(kv.first);
        for (size_t a = 0; a < kv.second.legal_actions.size(); ++a) {
          kv.second.cumulative_regrets[a] = fz == frozen.end() ? ExternalSamplingMCCFRSolver::kInitialTableValues
                                                               : fz->second.cumulative_regrets[a];
          kv.second.cumulative_policy[a] = fz == frozen.end() ? ExternalSamplingMCCFRSolver::kInitialTableValues
                                                              : fz->second.cumulative_policy[a];
        }
      }
    }
    for (auto& kv : table) {
      const auto& d = delta.at(kv.first);
      for (size_t a = 0; a < kv.second.legal_actions.size(); ++a) {
        kv.second.cumulative_regrets[a] += d.cumulative_regrets[a];
        kv.second.cumulative_policy[a] += d.cumulative_policy[a];
      }
    }
    return 0;
  });
#endif
}
// A FULL-SIZE mini-batch of the device's external-sampling MCCFR on the CPU, in seconds instead of hours: the summed
// regret / average-policy increments of trajectories [first, first + count) against ONE frozen table, `threads`
// trajectories at a time.  osgo_mccfr_minibatch above copies the whole table per trajectory (fine for 500
// trajectories, hopeless for 2^20); here the traversal of external_sampling_mccfr.cc:122-186 (AverageType::kSimple) is
// written against the frozen rows directly — regrets READ from the caller's table (a row the caller does not list
// counts as its initial 1e-6, external_sampling_mccfr.h:59), increments ADDED to a per-thread delta table — which is the
// same function of (table, uniforms) because a traversal never meets an infostate twice (perfect recall: the key holds
// the traverser's own actions).  Everything game-side is the library's own State (Child, LegalActions, ChanceOutcomes,
// InformationStateString, PlayerReturn), SampleAction (spiel.cc:372-409) and CFRInfoStateValues::ApplyRegretMatching /
// SampleActionIndex (cfr.cc:596-628), so in the -DOSGO_GENUINE_REFERENCE build the rules, keys and regret matching are
// the reference's own code.  Streams: the device's (see osgo_mccfr_minibatch).  Per-thread sums are added in thread
// order; `mass` returns, per row and action, the sum of |regret increment| — the scale a summation-order tolerance is
// stated against.  keys: n_rows newline-separated infostate strings; regrets / out arrays are [n_rows, amax] row-major.
namespace {
struct FrozenReplay {
  const Game* game = nullptr;
  int P = 0, amax = 0;
  const std::unordered_map<std::string, int>* index = nullptr;
  const double* regrets = nullptr;
  std::vector<double> d_reg, d_cum, mass;
  std::vector<int64_t> visits;
  CounterRng rng{0};
  uint64_t s0 = 0;
  double Walk(const State& state, Player player, int depth, int b1) {
    if (state.IsTerminal()) return state.PlayerReturn(player);                    // :125-127
    if (state.IsChanceNode()) {                                                   // :127-131
      const Action a = SampleAction(state.ChanceOutcomes(), rng.Unit()).first;
      return Walk(*state.Child(a), player, depth, b1);
    }
    const Player cur = state.CurrentPlayer();                                     // :132-146
    const std::string key = state.InformationStateString(cur);
    const std::vector<Action> legal = state.LegalActions();
    const int n = static_cast<int>(legal.size());
    CFRInfoStateValues copy(legal, ExternalSamplingMCCFRSolver::kInitialTableValues);
    auto it = index->find(key);
    const int row = it == index->end() ? -1 : it->second;
    if (row >= 0)
      for (int a = 0; a < n; ++a) copy.cumulative_regrets[a] = regrets[static_cast<size_t>(row) * amax + a];
    copy.ApplyRegretMatching();
    double value = 0;
    std::vector<double> child_values(n, 0.0);
    if (cur != player) {                                                          // :150-154
      const int a = copy.SampleActionIndex(0.0, rng.Unit());
      value = Walk(*state.Child(legal[a]), player, depth, b1);
    } else {                                                                      // :155-162
      for (int a = 0; a < n; ++a) {
        if (depth == 0) rng.s = s0 + (1 + static_cast<uint64_t>(a)) * 0xD6E8FEB86659FD93ULL;
        if (depth == 1) rng.s = s0 + (16 + 8 * static_cast<uint64_t>(b1) + static_cast<uint64_t>(a)) * 0xD6E8FEB86659FD93ULL;
        child_values[a] = Walk(*state.Child(legal[a]), player, depth + 1, depth == 0 ? a : b1);
        value += copy.current_policy[a] * child_values[a];
      }
    }
    if (row < 0) throw std::runtime_error("osgo_mccfr_frozen_replay: infostate not in the caller's table: " + key);
    ++visits[row];
    if (cur == player)                                                            // :164-172
      for (int a = 0; a < n; ++a) {
        const double term = child_values[a] - value;
        d_reg[static_cast<size_t>(row) * amax + a] += term;
        mass[static_cast<size_t>(row) * amax + a] += std::fabs(term);
      }
    if (cur == ((player + 1) % P))                                                // :177-183 (kSimple)
      for (int a = 0; a < n; ++a) d_cum[static_cast<size_t>(row) * amax + a] += copy.current_policy[a];
    return value;
  }
};
}  // namespace

int osgo_mccfr_frozen_replay(void* g, uint64_t seed, int64_t first, int64_t count, int threads, int n_rows, int amax,
                             const char* keys, const double* regrets, double* d_regrets, double* d_cum, double* mass,
                             int64_t* visits) {
  return Guard([&] {
    ORACLE_CHECK(g && keys && regrets && d_regrets && d_cum && mass && visits && threads >= 1 && count >= 0);
    const std::string game_string = static_cast<GameH*>(g)->game->ToString();
    std::unordered_map<std::string, int> index;
    {
      const char* p = keys;
      for (int r = 0; r < n_rows; ++r) {
        const char* e = std::strchr(p, '\n');
        std::string k = e ? std::string(p, e) : std::string(p);
        index.emplace(std::move(k), r);
        p = e ? e + 1 : p + std::strlen(p);
      }
      ORACLE_CHECK(static_cast<int>(index.size()) == n_rows);
    }
    const size_t cells = static_cast<size_t>(n_rows) * amax;
    std::vector<FrozenReplay> w(threads);
    std::vector<std::shared_ptr<const Game>> games(threads);  // one Game per thread (see osgo_bench_env_steps)
    std::vector<std::string> errors(threads);
    std::vector<std::thread> workers;
    for (int t = 0; t < threads; ++t) {
      games[t] = LoadGame(game_string);
      FrozenReplay& r = w[t];
      r.game = games[t].get(); r.P = games[t]->NumPlayers(); r.amax = amax; r.index = &index; r.regrets = regrets;
      r.d_reg.assign(cells, 0.0); r.d_cum.assign(cells, 0.0); r.mass.assign(cells, 0.0); r.visits.assign(n_rows, 0);
      const int64_t lo = first + count * t / threads, hi = first + count * (t + 1) / threads;
      workers.emplace_back([&r, &errors, t, lo, hi, seed] {
        try {
          for (int64_t traj = lo; traj < hi; ++traj) {
            r.rng = CounterRng(seed, static_cast<uint64_t>(traj), 0);
            r.s0 = r.rng.s;
            r.Walk(*r.game->NewInitialState(), static_cast<Player>(traj % r.P), 0, 0);
          }
        } catch (const std::exception& e) { errors[t] = e.what(); }
      });
    }
    for (auto& t : workers) t.join();
    for (const std::string& e : errors) if (!e.empty()) throw std::runtime_error(e);
    std::fill(d_regrets, d_regrets + cells, 0.0);
    std::fill(d_cum, d_cum + cells, 0.0);
    std::fill(mass, mass + cells, 0.0);
    std::fill(visits, visits + n_rows, 0);
    for (int t = 0; t < threads; ++t) {
      for (size_t k = 0; k < cells; ++k) { d_regrets[k] += w[t].d_reg[k]; d_cum[k] += w[t].d_cum[k]; mass[k] += w[t].mass[k]; }
      for (int r = 0; r < n_rows; ++r) visits[r] += w[t].visits[r];
    }
    return 0;
  });
}
// ExternalSamplingMCCFRSolver::FullUpdateAverage (external_sampling_mccfr.cc:188-231) on the table as it is: the
// second half of a kFull RunIteration, so that the device's mini-batch + full-average schedule can be replayed.
int osgo_mccfr_full_average(void* h) {
#ifdef OSGO_GENUINE_REFERENCE
  (void)h;
  g_err = "FullUpdateAverage on its own is a hook of the restatement (it is private in the reference)";
  return -1;
#else
  return Guard([&] {
    auto* c = static_cast<CfrH*>(h);
    ORACLE_CHECK(c->mccfr != nullptr);
    c->mccfr->FullUpdateAverageFromRoot();
    return 0;
  });
#endif
}
// The reference's text checkpoint (cfr.cc:284-307 CFRSolverBase::Serialize; DeserializeCFRSolver /
// DeserializeCFRPlusSolver cfr.cc:699-723).  Genuine build only: the restatement does not restate the
// wire format (the product's host mirror does, open_spiel_amd/csrc/host/osg_spiel.h; these two entry
// points let the tests load its checkpoints into the real reference and vice versa).
int osgo_cfr_serialize(void* h, int double_precision, char* buf, int cap) {
#ifdef OSGO_GENUINE_REFERENCE
  return Guard([&] {
    auto* c = static_cast<CfrH*>(h);
    if (c->mccfr) return CopyStr(c->mccfr->Serialize(double_precision), buf, cap);       // external_sampling_mccfr.cc:82-120
    if (c->osmccfr) return CopyStr(c->osmccfr->Serialize(double_precision), buf, cap);   // outcome_sampling_mccfr.cc:76-112
    ORACLE_CHECK(c->cfr);
    return CopyStr(c->cfr->Serialize(double_precision), buf, cap);
  });
#else
  (void)h; (void)double_precision; (void)buf; (void)cap;
  g_err = "the checkpoint format is only available from the genuine reference build";
  return -1;
#endif
}
// kind: 0 CFRSolver, 1 CFRPlusSolver, 2 / 3 ExternalSamplingMCCFRSolver, 5 OutcomeSamplingMCCFRSolver.
void* osgo_cfr_deserialize(const char* text, int kind) {
#ifdef OSGO_GENUINE_REFERENCE
  try {
    auto* h = new CfrH;
    const std::string serialized(text);
    h->game = PartiallyDeserializeCFRSolver(serialized).game;
    if (kind == 0) h->cfr = DeserializeCFRSolver(serialized);
    else if (kind == 2 || kind == 3) h->mccfr = DeserializeExternalSamplingMCCFRSolver(serialized);
    else if (kind == 5) h->osmccfr = DeserializeOutcomeSamplingMCCFRSolver(serialized);
    else h->cfr = DeserializeCFRPlusSolver(serialized);
    return h;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
#else
  (void)text; (void)kind;
  g_err = "the checkpoint format is only available from the genuine reference build";
  return nullptr;
#endif
}
int osgo_cfr_num_infostates(void* h) {
  return static_cast<int>(static_cast<CfrH*>(h)->Table().size());
}
// Rows sorted by key.  keys: '\n'-joined.  Arrays are [I, amax], padded with 0;
// nact[I]; legal [I, amax] padded with -1.
int osgo_cfr_tables(void* h, int amax, char* keys, int keys_cap, int* nact,
                    int64_t* legal, double* regrets, double* cum_policy,
                    double* cur_policy, double* avg_policy) {
  return Guard([&] {
    auto& table = static_cast<CfrH*>(h)->Table();
    std::vector<std::string> ks;
    for (const auto& kv : table) ks.push_back(kv.first);
    std::sort(ks.begin(), ks.end());
    std::string joined;
    for (size_t i = 0; i < ks.size(); ++i) {
      if (i) joined += "\n";
      joined += ks[i];
    }
    CopyStr(joined, keys, keys_cap);
    for (size_t i = 0; i < ks.size(); ++i) {
      const CFRInfoStateValues& v = table.at(ks[i]);
      ActionsAndProbs avg = AverageFromValues(v);
      nact[i] = v.num_actions();
      for (int a = 0; a < amax; ++a) {
        bool in = a < v.num_actions();
        legal[i * amax + a] = in ? v.legal_actions[a] : -1;
        regrets[i * amax + a] = in ? v.cumulative_regrets[a] : 0;
        cum_policy[i * amax + a] = in ? v.cumulative_policy[a] : 0;
        cur_policy[i * amax + a] = in ? v.current_policy[a] : 0;
        avg_policy[i * amax + a] = in ? avg[a].second : 0;
      }
    }
    return static_cast<int>(joined.size());
  });
}
// which: 0 NashConv(average), 1 Exploitability(average), 2 NashConv(current)
int osgo_cfr_eval(void* h, int which, double* out) {
  return Guard([&] {
    auto* c = static_cast<CfrH*>(h);
    std::shared_ptr<Policy> pol;
    if (which == 2) {
      ORACLE_CHECK(c->cfr);
      pol = c->cfr->CurrentPolicy();
    } else {
      pol = c->Average();
    }
#ifdef OSGO_GENUINE_REFERENCE
    // The sampling solvers' average policy falls back to UniformPolicy for infostates the table
    // has not seen, which only answers GetStatePolicy(state): hence use_state_get_policy = true,
    // as in external_sampling_mccfr_test.cc:41.
    if (c->cfr) {
      *out = which == 1 ? Exploitability(*c->game, *pol) : NashConv(*c->game, *pol);
    } else {
      const double nc = NashConv(*c->game, *pol, /*use_state_get_policy=*/true);
      *out = which == 1 ? nc / c->game->NumPlayers() : nc;
    }
#else
    *out = which == 1 ? Exploitability(*c->game, *pol) : NashConv(*c->game, *pol);
#endif
    return 0;
  });
}
// Expected value of the average policy for each player (cfr_test.cc:36-62).
int osgo_cfr_expected_returns(void* h, double* out) {
  return Guard([&] {
    auto* c = static_cast<CfrH*>(h);
    std::shared_ptr<Policy> pol = c->Average();
    std::vector<double> v = ExpReturns(*c->game->NewInitialState(), *pol);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return 0;
  });
}

// Value of `player`'s best response to the average policy at the root: TabularBestResponse(game, player, policy)
// .Value(root) (best_response.h:57-72, best_response.cc:194-227) — one of the P terms NashConv sums
// (tabular_exploitability.cc:77-89), so a large tree can be checked one player at a time.
int osgo_cfr_best_response_value(void* h, int player, double* out) {
  return Guard([&] {
    auto* c = static_cast<CfrH*>(h);
    std::shared_ptr<Policy> pol = c->Average();
#ifdef OSGO_GENUINE_REFERENCE
    open_spiel::algorithms::TabularBestResponse br(*c->game, player, pol.get());
    *out = br.Value(*c->game->NewInitialState());
#else
    *out = BestResponseValue(*c->game, player, *pol);
#endif
    return 0;
  });
}

// Judge an externally supplied tabular policy (e.g. downloaded from the GPU):
// keys '\n'-joined, rows [I, amax] of (action, prob), nact[I].
// which: 0 NashConv, 1 Exploitability; also writes expected returns to ev[P].
int osgo_eval_policy(void* g, const char* keys, int amax, const int* nact,
                     const int64_t* actions, const double* probs, int which,
                     double* out, double* ev) {
  return Guard([&] {
    const Game& game = *static_cast<GameH*>(g)->game;
    TabularPolicy pol;
    std::string all(keys);
    size_t pos = 0;
    int row = 0;
    while (pos <= all.size()) {
      size_t nl = all.find('\n', pos);
      if (nl == std::string::npos) nl = all.size();
      std::string key = all.substr(pos, nl - pos);
      ActionsAndProbs ap;
      for (int a = 0; a < nact[row]; ++a)
        ap.push_back({actions[row * amax + a], probs[row * amax + a]});
      SetRow(&pol, key, ap);
      ++row;
      pos = nl + 1;
      if (nl == all.size()) break;
    }
    *out = which == 1 ? Exploitability(game, pol) : NashConv(game, pol);
    if (ev) {
      std::vector<double> v = ExpReturns(*game.NewInitialState(), pol);
      for (size_t i = 0; i < v.size(); ++i) ev[i] = v[i];
    }
    return 0;
  });
}
// Named policies: 0 uniform, 1 first-action, 2 kuhn optimal(alpha).
int osgo_eval_named_policy(void* g, int which_policy, double alpha, int which,
                           double* out) {
  return Guard([&] {
    const Game& game = *static_cast<GameH*>(g)->game;
    TabularPolicy pol = which_policy == 0   ? GetUniformPolicy(game)
                        : which_policy == 1 ? GetFirstActionPolicy(game)
                                            : KuhnOptimal(alpha);
    *out = which == 1 ? Exploitability(game, pol) : NashConv(game, pol);
    return 0;
  });
}

// Tree census (integration_tests/api_test.py:75-101): out = chance, decision,
// terminal node counts and the number of distinct infostates.
static void Census(const State& s, int64_t* out, std::unordered_map<std::string, int>* seen) {
  if (s.IsTerminal()) {
    ++out[2];
    return;
  }
  if (s.IsChanceNode()) {
    ++out[0];
    for (const auto& ap : s.ChanceOutcomes()) Census(*s.Child(ap.first), out, seen);
    return;
  }
  ++out[1];
  (*seen)[s.InformationStateString()] = 1;
  for (Action a : s.LegalActions()) Census(*s.Child(a), out, seen);
}
int osgo_tree_census(void* g, int64_t* out) {
  return Guard([&] {
    const Game& game = *static_cast<GameH*>(g)->game;
    out[0] = out[1] = out[2] = 0;
    std::unordered_map<std::string, int> seen;
    Census(*game.NewInitialState(), out, &seen);
    out[3] = static_cast<int64_t>(seen.size());
    return 0;
  });
}

// ---------------------------------------------------------------------------
// cpu_baseline timing legs (bench.py).  Each returns elapsed seconds via *secs
// and the number of units processed via *units; `threads` independent workers.
// ---------------------------------------------------------------------------
// (a) env steps: per state LegalActions() + ApplyAction() + IsTerminal() +
//     Returns() + CurrentPlayer() over pre-generated states (Clone()d from a
//     pool of `pool` positions reached by random play, depth = hash mod 36,
//     non-terminal) applying one seeded random legal action each.  Units =
//     env steps.  The pool is rebuilt outside the timed region.
int osgo_bench_env_steps(void* g, uint64_t seed, int64_t pool, int64_t total_steps,
                         int threads, double* secs, int64_t* units) {
  return Guard([&] {
    // One Game object per worker thread, as separate processes would have: every State holds a
    // shared_ptr to its Game (spiel.h:906), and Clone() from many threads on ONE Game would
    // serialise on that reference count instead of measuring the game logic.
    const std::string game_string = static_cast<GameH*>(g)->game->ToString();
    const int64_t per_thread = std::max<int64_t>(pool / threads, 1);
    std::vector<std::shared_ptr<const Game>> games(threads);
    std::vector<std::vector<std::unique_ptr<State>>> states(threads);
    std::vector<std::vector<Action>> actions(threads);
    for (int w = 0; w < threads; ++w) {
      games[w] = LoadGame(game_string);
      states[w].resize(per_thread);
      actions[w].resize(per_thread);
      for (int64_t j = 0; j < per_thread; ++j) {
        CounterRng rng(seed, static_cast<uint64_t>(w * per_thread + j));
        for (;;) {
          std::unique_ptr<State> s = games[w]->NewInitialState();
          int depth = static_cast<int>(rng.Below(36));
          for (int t = 0; t < depth && !s->IsTerminal(); ++t) {
            if (s->IsChanceNode()) {
              s->ApplyAction(SampleAction(s->ChanceOutcomes(), rng.Unit()).first);
            } else {
              std::vector<Action> la = s->LegalActions();
              s->ApplyAction(la[rng.Below(static_cast<uint32_t>(la.size()))]);
            }
          }
          if (s->IsTerminal()) continue;
          std::vector<Action> la = s->LegalActions();
          actions[w][j] = la[rng.Below(static_cast<uint32_t>(la.size()))];
          states[w][j] = std::move(s);
          break;
        }
      }
    }
    std::vector<int64_t> done(threads, 0);
    std::vector<double> sink(threads, 0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int w = 0; w < threads; ++w) {
      workers.emplace_back([&, w] {
        int64_t quota = total_steps / threads;
        double acc = 0;
        for (int64_t k = 0; k < quota; ++k) {
          const int64_t i = k % per_thread;
          std::unique_ptr<State> s = states[w][i]->Clone();
          std::vector<Action> la = s->LegalActions();
          s->ApplyAction(actions[w][i]);
          acc += la.size() + s->IsTerminal() + s->Returns()[0] + s->CurrentPlayer();
          acc += s->LegalActions().size();
        }
        done[w] = quota;
        sink[w] = acc;
      });
    }
    for (auto& t : workers) t.join();
    auto t1 = std::chrono::steady_clock::now();
    *secs = std::chrono::duration<double>(t1 - t0).count();
    *units = 0;
    for (int64_t d : done) *units += d;
    if (sink[0] == 1234.5678) g_err = "sink";
    return 0;
  });
}
// (a2) tensor pack: State::ObservationTensor(player) into a preallocated span (spiel.h:713-714) for the
//      player to move, over a pool of seeded non-terminal positions; units = tensors written.
int osgo_bench_observation(void* g, uint64_t seed, int64_t pool, int64_t total, int threads, double* secs,
                           int64_t* units) {
  return Guard([&] {
    const std::string game_string = static_cast<GameH*>(g)->game->ToString();
    const int64_t per_thread = std::max<int64_t>(pool / threads, 1);
    std::vector<std::shared_ptr<const Game>> games(threads);  // one Game per thread (see osgo_bench_env_steps)
    std::vector<std::vector<std::unique_ptr<State>>> states(threads);
    for (int w = 0; w < threads; ++w) {
      games[w] = LoadGame(game_string);
      states[w].resize(per_thread);
      for (int64_t j = 0; j < per_thread; ++j) {
        CounterRng rng(seed, static_cast<uint64_t>(w * per_thread + j));
        for (;;) {
          std::unique_ptr<State> s = games[w]->NewInitialState();
          int depth = static_cast<int>(rng.Below(36));
          for (int t = 0; t < depth && !s->IsTerminal(); ++t) {
            if (s->IsChanceNode()) {
              s->ApplyAction(SampleAction(s->ChanceOutcomes(), rng.Unit()).first);
            } else {
              std::vector<Action> la = s->LegalActions();
              s->ApplyAction(la[rng.Below(static_cast<uint32_t>(la.size()))]);
            }
          }
          if (s->IsTerminal() || s->IsChanceNode()) continue;
          states[w][j] = std::move(s);
          break;
        }
      }
    }
    const int size = games[0]->ObservationTensorSize();
    std::vector<double> sink(threads, 0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int w = 0; w < threads; ++w) {
      workers.emplace_back([&, w] {
        std::vector<float> out(size);
        double acc = 0;
        const int64_t quota = total / threads;
        for (int64_t k = 0; k < quota; ++k) {
          const State& s = *states[w][k % per_thread];
          WriteTensor(s, 0, s.CurrentPlayer(), out.data(), size);
          acc += out[k % size];
        }
        sink[w] = acc;
      });
    }
    for (auto& t : workers) t.join();
    auto t1 = std::chrono::steady_clock::now();
    *secs = std::chrono::duration<double>(t1 - t0).count();
    *units = (total / threads) * threads;
    if (sink[0] == 1234.5678) g_err = "sink";
    return 0;
  });
}
// (b) random playouts (benchmark_game.cc-equivalent): units = moves.
int osgo_bench_playouts(void* g, uint64_t seed, int64_t sims, int threads,
                        double* secs, int64_t* moves) {
  return Guard([&] {
    const std::string game_string = static_cast<GameH*>(g)->game->ToString();
    std::vector<std::shared_ptr<const Game>> games(threads);  // one Game per thread (see osgo_bench_env_steps)
    for (int w = 0; w < threads; ++w) games[w] = LoadGame(game_string);
    std::vector<int64_t> done(threads, 0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int w = 0; w < threads; ++w) {
      workers.emplace_back([&, w] {
        int64_t count = 0;
        const Game& game = *games[w];
        for (int64_t k = w; k < sims; k += threads) {
          CounterRng rng(seed, static_cast<uint64_t>(k));
          std::unique_ptr<State> s = game.NewInitialState();
          while (!s->IsTerminal()) {
            if (s->IsChanceNode()) {
              s->ApplyAction(SampleAction(s->ChanceOutcomes(), rng.Unit()).first);
            } else {
              std::vector<Action> la = s->LegalActions();
              s->ApplyAction(la[rng.Below(static_cast<uint32_t>(la.size()))]);
            }
            ++count;
          }
        }
        done[w] = count;
      });
    }
    for (auto& t : workers) t.join();
    auto t1 = std::chrono::steady_clock::now();
    *secs = std::chrono::duration<double>(t1 - t0).count();
    *moves = 0;
    for (int64_t d : done) *moves += d;
    return 0;
  });
}
// (c) MCTS sims/s: `roots` independent MCTSBot searches from positions reached
// by hash(i) mod depth_mod random moves; units = simulations.
int osgo_bench_mcts(void* g, uint64_t seed, int roots, int depth_mod,
                    int max_simulations, int n_rollouts, double uct_c, int threads,
                    double* secs, int64_t* sims) {
  return Guard([&] {
    const std::string game_string = static_cast<GameH*>(g)->game->ToString();
    std::vector<std::shared_ptr<const Game>> games(threads);  // one Game per thread (see osgo_bench_env_steps)
    for (int w = 0; w < threads; ++w) games[w] = LoadGame(game_string);
    std::vector<std::unique_ptr<State>> starts(roots);
    for (int i = 0; i < roots; ++i) {
      const Game& game = *games[i % threads];  // root i is searched by thread i % threads
      CounterRng rng(seed, static_cast<uint64_t>(i));
      for (;;) {
        std::unique_ptr<State> s = game.NewInitialState();
        int depth = depth_mod > 0 ? static_cast<int>(rng.Below(depth_mod)) : 0;
        for (int t = 0; t < depth && !s->IsTerminal(); ++t) {
          std::vector<Action> la = s->LegalActions();
          s->ApplyAction(la[rng.Below(static_cast<uint32_t>(la.size()))]);
        }
        if (s->IsTerminal()) continue;
        starts[i] = std::move(s);
        break;
      }
    }
    std::vector<int64_t> done(threads, 0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int w = 0; w < threads; ++w) {
      workers.emplace_back([&, w] {
        int64_t count = 0;
        for (int i = w; i < roots; i += threads) {
          auto ev = std::make_shared<RandomRolloutEvaluator>(n_rollouts, 42 + i);
          MCTSBot bot(*games[w], ev, uct_c, max_simulations, 1000, false, 42 + i, false);
          std::unique_ptr<SearchNode> root = bot.MCTSearch(*starts[i]);
          count += root->explore_count;
        }
        done[w] = count;
      });
    }
    for (auto& t : workers) t.join();
    auto t1 = std::chrono::steady_clock::now();
    *secs = std::chrono::duration<double>(t1 - t0).count();
    *sims = 0;
    for (int64_t d : done) *sims += d;
    return 0;
  });
}
// (c') BASELINE.json configs[0], the reference's own mcts_test.cc:35-49 setup: tic_tac_toe
// MCTSBot(RandomRolloutEvaluator(20, 42), uct_c = 2, max_simulations, max_memory_mb = 5, solve = true, seed 42,
// verbose = false), MCTSearch from the initial state and from the three MCTS-Solver positions of
// mcts_test.cc:126-155, `repeats` times; units = simulations (root explore_count; a solved root stops early).
int osgo_bench_mcts_config1(void* g, int max_simulations, int repeats, double* secs, int64_t* sims) {
  return Guard([&] {
    const Game& game = *static_cast<GameH*>(g)->game;
    const char* lines[4] = {"", "x(1,1) o(0,0) x(2,2)", "x(1,1) o(0,0) x(2,2) o(0,1) x(0,2)", "x(0,1) o(2,2)"};
    std::vector<std::unique_ptr<State>> starts;
    for (const char* line : lines) {
      std::unique_ptr<State> s = game.NewInitialState();
      std::string rest = line;
      while (!rest.empty()) {
        const size_t sp = rest.find(' ');
        const std::string tok = rest.substr(0, sp);
        rest = sp == std::string::npos ? "" : rest.substr(sp + 1);
        bool found = false;
        for (Action a : s->LegalActions())
          if (s->ActionToString(s->CurrentPlayer(), a) == tok) { s->ApplyAction(a); found = true; break; }
        if (!found) throw std::runtime_error("osgo_bench_mcts_config1: no legal action " + tok);
      }
      starts.push_back(std::move(s));
    }
    int64_t count = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < repeats; ++rep) {
      for (const auto& start : starts) {
        auto ev = std::make_shared<RandomRolloutEvaluator>(20, 42);
        MCTSBot bot(game, ev, 2.0, max_simulations, 5, true, 42, false);
        std::unique_ptr<SearchNode> root = bot.MCTSearch(*start);
        count += root->explore_count;
      }
    }
    auto t1 = std::chrono::steady_clock::now();
    *secs = std::chrono::duration<double>(t1 - t0).count();
    *sims = count;
    return 0;
  });
}
// (d) solver iterations/s (kind as in osgo_cfr_create); one solver per thread.
int osgo_bench_cfr(void* g, int kind, int iters, int threads, double* secs) {
  return Guard([&] {
    const std::string game_string = static_cast<GameH*>(g)->game->ToString();
    std::vector<void*> gs(threads), hs(threads);
    for (int w = 0; w < threads; ++w) {  // one Game per thread (see osgo_bench_env_steps)
      gs[w] = osgo_load_game(game_string.c_str());
      hs[w] = osgo_cfr_create(gs[w], kind, 1 + w);
    }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int w = 0; w < threads; ++w)
      workers.emplace_back([&, w] { osgo_cfr_iterate(hs[w], iters); });
    for (auto& t : workers) t.join();
    auto t1 = std::chrono::steady_clock::now();
    *secs = std::chrono::duration<double>(t1 - t0).count();
    for (void* h : hs) osgo_cfr_free(h);
    for (void* gg : gs) osgo_free_game(gg);
    return 0;
  });
}

}  // extern "C"
