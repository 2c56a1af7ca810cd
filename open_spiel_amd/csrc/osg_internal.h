// Library-internal declarations shared by the .hip translation units.
#ifndef OSG_INTERNAL_H_
#define OSG_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/osg_abi.h"
#include "osg_common.h"
#include "osg_game_boards.h"
#include "osg_game_poker.h"
#include "osg_sample.h"

namespace osg {

enum GameKind { kTtt = 0, kC4 = 1, kHex = 2, kKuhn = 3, kLeduc = 4 };

// A parsed, validated game: description + the device parameter block.
struct GameSpec {
  osg_game_desc desc;
  int hex_nw = 0;  // u32 words per hex bit plane: 1..4 (boards of up to 128 actions), 6 / 8 / 12 (up to 19 x 19)
  bool hex_fold = false;  // cells <= 32 * hex_nw - 5: the meta word folded into the planes' spare bits (a 4 * hex_nw word record)
  bool c4_std = false;  // connect_four with the default 6x7x4 geometry (constant-folded kernels)
  bool c4_wide = false;  // connect_four above 64 board bits: two plane words per colour (C4Wide)
  bool leduc_big = false;  // leduc_poker with 4 to 10 players: the five-plane record (LeducBig)
  bool hex_explicit = false;  // hex(string_rep=explicit): edge-connection glyphs in the board string
  Ttt::Params ttt;
  C4::Params c4;
  HexT<1>::Params hex1;
  HexT<2>::Params hex2;
  HexT<3>::Params hex3;
  HexT<4>::Params hex4;
  HexT<6>::Params hex6;    // 13 x 13 (169 cells)
  HexT<8>::Params hex8;    // 15 x 15 (225 cells)
  HexT<12>::Params hex12;  // 19 x 19 (361 cells)
  Kuhn::Params kuhn;
  Leduc::Params leduc;
  std::map<std::string, std::string> params;  // as given + defaults (strings)
};

int set_error(int code, const std::string& msg);

// hipFuncAttributeMaxDynamicSharedMemorySize is one value per KERNEL, not per solver / tree: a second user with a
// smaller footprint must not lower the cap under a first one that is still in use.  Raises only; asked of the runtime
// once per (device, kernel) and size step, not per launch.
inline hipError_t raise_lds_cap(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::unordered_map<std::string, int> cap;   // per (device, kernel)
  int device = 0;
  (void)hipGetDevice(&device);
  std::lock_guard<std::mutex> lock(mu);
  int& have = cap[std::to_string(device) + ":" + std::to_string(reinterpret_cast<uintptr_t>(kernel))];
  if (bytes <= have) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

// Random playouts of hex on a one-row or one-column board never end: with the reference's `else if` between a
// colour's two edges (hex.cc:122-126,146-150) a stone on such a board can only ever carry ONE edge label, so that
// colour never wins and a filled board is a state that is not terminal and has no legal action.  (The reference's
// RandomRolloutEvaluator would index an empty LegalActions() there.)  Entry points that play out refuse these
// boards instead of hanging the device.
inline void hex_dims(const GameSpec& spec, int* rows, int* cols, int* cells) {
  switch (spec.hex_nw) {
    case 1: *rows = spec.hex1.rows; *cols = spec.hex1.cols; *cells = spec.hex1.cells; break;
    case 2: *rows = spec.hex2.rows; *cols = spec.hex2.cols; *cells = spec.hex2.cells; break;
    case 3: *rows = spec.hex3.rows; *cols = spec.hex3.cols; *cells = spec.hex3.cells; break;
    case 4: *rows = spec.hex4.rows; *cols = spec.hex4.cols; *cells = spec.hex4.cells; break;
    case 6: *rows = spec.hex6.rows; *cols = spec.hex6.cols; *cells = spec.hex6.cells; break;
    case 8: *rows = spec.hex8.rows; *cols = spec.hex8.cols; *cells = spec.hex8.cells; break;
    default: *rows = spec.hex12.rows; *cols = spec.hex12.cols; *cells = spec.hex12.cells; break;
  }
}
inline int refuse_endless_playouts(const GameSpec& spec, const char* who) {
  if (spec.desc.game_kind != kHex) return OSG_OK;
  int rows = 0, cols = 0, cells = 0;
  hex_dims(spec, &rows, &cols, &cells);
  if (rows >= 2 && cols >= 2) return OSG_OK;
  return set_error(OSG_ERR_UNSUPPORTED, std::string(who) + ": hex on a board with a single row or column has states that are "
                   "neither terminal nor have a legal action (one colour can never win): playouts would not end");
}
int parse_game(const char* game_string, GameSpec* out);

}  // namespace osg

struct osg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  unsigned long long* d_illegal = nullptr;  // device counter of illegal applies
  void* d_scratch = nullptr;                // reusable staging buffer
  size_t scratch_bytes = 0;
  void* h_pinned = nullptr;
  size_t pinned_bytes = 0;
  void* d_mcts_pool = nullptr;              // MCTS node pool, grow-only, reused across searches
  size_t mcts_pool_bytes = 0;
  double* d_mcts_logs = nullptr;            // log(n) table shared with the host libm
  int mcts_logs_n = 0;
  int32_t* d_mcts_queue = nullptr;          // wave-per-root search as a work queue: [0] next ticket, [1..256] the cost
  int64_t mcts_queue_roots = 0;             // histogram / bucket offsets, then the root order [roots] (grow-only)
  int num_cus = 0;
  // Lifetime: the creator holds one reference, every batch / solver / communicator made on the context
  // another; osg_ctx_destroy drops the creator's, and the device resources go with the last one, so a
  // batch destroyed after its context (garbage-collection order in a binding) never touches freed memory.
  int refs = 1;
  bool closed = false;
};
namespace osg {
void ctx_retain(osg_ctx* ctx);
void ctx_release(osg_ctx* ctx);
}

struct osg_batch {
  osg_ctx* ctx = nullptr;
  osg::GameSpec spec;
  int64_t n = 0;
  void* d_words = nullptr;  // state_words planes of n elements
  size_t bytes = 0;
};

#define OSG_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e__ = (call);                                                       \
    if (e__ != hipSuccess)                                                         \
      return osg::set_error(OSG_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

// Dispatch a generic lambda-like macro body over the concrete game type.
// Inside the body: `G` is the game struct and `P` its Params instance.
// OSG_DISPATCH serves the games whose legal mask is the engine's 4-word Mask and whose record is the two-word one (the
// wave-per-root search and the solvers' tree builder); OSG_DISPATCH_WIDE adds the hex boards above 128 actions,
// connect_four above 64 board bits and leduc_poker with 4+ players (the batch entry points of osg_kernels.hip — states,
// masks, steps, tensors, random steps, rollouts, environment steps — and the lane-per-root searches of osg_mcts.hip /
// osg_mcts_step.hip).
#define OSG_DISPATCH(spec, ...)                                                   \
  do {                                                                             \
    switch ((spec).desc.game_kind) {                                               \
      case osg::kTtt: { using G = osg::Ttt; const G::Params& P = (spec).ttt; __VA_ARGS__; } break; \
      case osg::kC4:                                                               \
        if ((spec).c4_std) { using G = osg::C4Std; const G::Params& P = (spec).c4; __VA_ARGS__; } \
        else if ((spec).c4_wide) return osg::set_error(OSG_ERR_UNSUPPORTED, "connect_four boards above 64 bits are served by " \
                                                       "the batch entry points and the lane-per-root searches, not by this one"); \
        else { using G = osg::C4; const G::Params& P = (spec).c4; __VA_ARGS__; }   \
        break;                                                                     \
      case osg::kKuhn: { using G = osg::Kuhn; const G::Params& P = (spec).kuhn; __VA_ARGS__; } break; \
      case osg::kLeduc:                                                            \
        if ((spec).leduc_big) return osg::set_error(OSG_ERR_UNSUPPORTED, "leduc_poker with more than 3 players is served by the " \
                                                    "batch entry points and the lane-per-root searches, not by this one"); \
        { using G = osg::Leduc; const G::Params& P = (spec).leduc; __VA_ARGS__; } break; \
      case osg::kHex:                                                              \
        switch ((spec).hex_nw) {                                                   \
          case 1: { using G = osg::HexT<1>; const G::Params& P = (spec).hex1; __VA_ARGS__; } break; \
          case 2: { using G = osg::HexT<2>; const G::Params& P = (spec).hex2; __VA_ARGS__; } break; \
          case 3: { using G = osg::HexT<3>; const G::Params& P = (spec).hex3; __VA_ARGS__; } break; \
          case 4: { using G = osg::HexT<4>; const G::Params& P = (spec).hex4; __VA_ARGS__; } break; \
          default: return osg::set_error(OSG_ERR_UNSUPPORTED, "hex boards above 128 actions are served by the batch entry " \
                                         "points and the lane-per-root searches, not by this one"); \
        }                                                                          \
        break;                                                                     \
      default: return osg::set_error(OSG_ERR_INVALID, "bad game kind");            \
    }                                                                              \
  } while (0)
#define OSG_DISPATCH_WIDE(spec, ...)                                              \
  do {                                                                             \
    switch ((spec).desc.game_kind) {                                               \
      case osg::kTtt: { using G = osg::Ttt; const G::Params& P = (spec).ttt; __VA_ARGS__; } break; \
      case osg::kC4:                                                               \
        if ((spec).c4_std) { using G = osg::C4Std; const G::Params& P = (spec).c4; __VA_ARGS__; } \
        else if ((spec).c4_wide) { using G = osg::C4Wide; const G::Params& P = (spec).c4; __VA_ARGS__; } \
        else { using G = osg::C4; const G::Params& P = (spec).c4; __VA_ARGS__; }   \
        break;                                                                     \
      case osg::kKuhn: { using G = osg::Kuhn; const G::Params& P = (spec).kuhn; __VA_ARGS__; } break; \
      case osg::kLeduc:                                                            \
        if ((spec).leduc_big) { using G = osg::LeducBig; const G::Params& P = (spec).leduc; __VA_ARGS__; } \
        else { using G = osg::Leduc; const G::Params& P = (spec).leduc; __VA_ARGS__; } \
        break;                                                                     \
      case osg::kHex:                                                              \
        switch ((spec).hex_nw) {                                                   \
          case 1: { using G = osg::HexT<1>; const G::Params& P = (spec).hex1; __VA_ARGS__; } break; \
          case 2: { using G = osg::HexT<2>; const G::Params& P = (spec).hex2; __VA_ARGS__; } break; \
          case 3: { using G = osg::HexT<3>; const G::Params& P = (spec).hex3; __VA_ARGS__; } break; \
          case 4: { using G = osg::HexT<4>; const G::Params& P = (spec).hex4; __VA_ARGS__; } break; \
          case 6: { using G = osg::HexT<6>; const G::Params& P = (spec).hex6; __VA_ARGS__; } break; \
          case 8: { using G = osg::HexT<8>; const G::Params& P = (spec).hex8; __VA_ARGS__; } break; \
          default: { using G = osg::HexT<12>; const G::Params& P = (spec).hex12; __VA_ARGS__; } break; \
        }                                                                          \
        break;                                                                     \
      default: return osg::set_error(OSG_ERR_INVALID, "bad game kind");            \
    }                                                                              \
  } while (0)

// Scratch helpers (device + pinned host staging owned by the context).
int osg_ctx_scratch(osg_ctx* ctx, size_t bytes, void** out);
int osg_ctx_pinned(osg_ctx* ctx, size_t bytes, void** out);

#endif  // OSG_INTERNAL_H_
