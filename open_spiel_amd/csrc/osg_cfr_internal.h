// Shared declarations of the tabular-solver translation units (round 6: osg_cfr.hip split by kernel family):
//
//   osg_cfr.hip         the flattened tree (build_tree), the tables, the C-ABI entry points and their dispatch
//   osg_cfr_small.hip   k_cfr, k_cfr_small            one workgroup (kuhn_poker; any small tree)
//   osg_cfr_split.hip   k_cfr_split                   one workgroup per deal subtree (leduc_poker)
//   osg_cfr_sub.hip     k_cfr_sub, k_gcfr_*           one persistent cooperative launch / a launch per phase (3-player leduc)
//   osg_cfr_eval.hip    k_policy_eval, k_geval_*, k_eval_jobs     ExpectedReturns / TabularBestResponse / NashConv
//   osg_cfr_mccfr.hip   k_mccfr*, k_os_mccfr*, ...    external / outcome sampling MCCFR and their entry points
//
// Every kernel is launched from the translation unit that defines it; what crosses the units are the device-side
// argument structs below, struct osg_cfr, and the host functions declared at the end.  Reference files: cfr.{h,cc},
// external_sampling_mccfr.{h,cc}, outcome_sampling_mccfr.cc, cfr_br.cc, expected_returns.cc, best_response.cc.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "osg_internal.h"

using namespace osg;

namespace osg_cfr_impl {

constexpr int kMaxA = 4;        // widest decision node the MCCFR frame holds (kuhn 2, leduc 3)
constexpr int kMaxPolicyRow = 8;  // widest policy row a thread regret-matches in registers (kuhn 2, leduc 3)
constexpr int kMaxFrames = 24;  // traverser decision nodes on one path
#ifndef OSG_MCCFR_FRAMES2
#define OSG_MCCFR_FRAMES2 1     // the flat ES-MCCFR kernel keeps the two upper frames of the traverser's stack in registers (0: A/B)
#endif
#ifndef OSG_MCCFR_PEEK
#define OSG_MCCFR_PEEK 0        // 1: the flat ES-MCCFR kernel forms the next draw while the node record is in flight — measured
                                // 2.7 % SLOWER (profiles/r05_ab_solvers.txt: the draws of traverser / terminal visits are wasted
                                // vector work on a SIMD that is already two thirds busy issuing); kept as a switch
#endif
constexpr double kMccfrInit = 0.000001;  // external_sampling_mccfr.h:59 kInitialTableValues

enum NodeKind : uint8_t { kChanceNode = 0, kDecisionNode = 1, kTerminalNode = 2 };

struct Tree {  // device pointers
  int H, I, A, P, D;
  const int32_t* level_off;    // [D+1]
  const int32_t* parent;       // [H]
  const int32_t* first_child;  // [H]
  const uint8_t* kind;         // [H]
  const uint8_t* nchild;       // [H]
  const uint8_t* aidx;         // [H] index of the incoming edge among the parent's children
  const int8_t* actor;         // [H] acting player, -1 at chance / terminal nodes
  const int32_t* info;         // [H] infostate id of a decision node, else -1
  const double* edge_prob;     // [H] chance probability of the incoming edge (parent is chance)
  const double* term_ret;      // [H, P] Returns() of terminal nodes
  const int32_t* mem_off;      // [I+1]
  const int32_t* mem;          // member histories of every infostate, DFS order
  const int32_t* nact;         // [I]
  const int8_t* info_player;   // [I]
};

struct Tables {  // [I, A] fp64
  double* regrets;
  double* cum;
  double* cur;
};

// ---------------------------------------------------------------------------
// CFRInfoStateValues::ApplyRegretMatching (cfr.cc:596-615) on one row.
// ---------------------------------------------------------------------------
OSG_D void regret_match_row(const double* regrets, double* policy, int n) {
  double sum_pos = 0.0;
  for (int a = 0; a < n; ++a)
    if (regrets[a] > 0) sum_pos += regrets[a];
  for (int a = 0; a < n; ++a) {
    if (sum_pos > 0) policy[a] = regrets[a] > 0 ? regrets[a] / sum_pos : 0.0;
    else policy[a] = 1.0 / n;
  }
}


struct SmallTree {  // device pointers to the extra host-built arrays
  const int32_t* path_off;    // [M+1] per decision history (member order)
  const int32_t* path;        // entries: slot << 24 | is_chance << 23 | index
  int M;                      // decision histories
  int n_path;
  int L0 = 0;                 // the first level that holds a decision history: the sweep of k_cfr_small stops there — the
                              // values of the chance levels above (the deals) are read by nobody (phase B reads a decision
                              // history's own value and its children's), and for kuhn_poker they were 2 of its 5 level steps
};

struct SmallGlobal {  // global-memory homes of the same arrays, for trees too big for LDS (leduc)
  double* value;         // [H, P]
  double* dreg;          // [M, A]
  double* dpol;          // [M, A]
  int32_t* skip;         // [M]
  const int32_t* meta;   // [H] kind | nchild << 2 | (actor + 1) << 10
  const int32_t* info_player;  // [I]
};

struct SplitTree {
  int G, L, NL, NM, NI;          // subtrees, cut level, padded histories / members / infostates per subtree
  const int32_t* nloc;           // [G] histories of the subtree
  const int32_t* hist_desc;      // [G, NL] kind | nchild << 2 | level << 10 | (actor + 1) << 16
  const int32_t* hist_fc;        // [G, NL] LOCAL index of the first child
  const int32_t* hist_row;       // [G, NL] info * A of a decision node
  const int32_t* hist_glob;      // [G, NL] the history's index in the whole tree
  const int32_t* mem_m;          // [G, NM] member index (position in Tree::mem), -1 = padding
  const int32_t* mem_hloc;       // [G, NM] its history, local index
  const int32_t* info_list;      // [G, NI] infostates with a member in the subtree, -1 = padding
  double* terms;                 // [2][M][kSplitRec]: buffer (pass parity) x {own reach or -1, A regret terms} per member
  unsigned int* bar;             // [0] arrival counter (zero between launches), [1] error flag, [2] sticky error, [3] exit counter
  unsigned int* host_err;        // pinned host word raised on a timeout: the host's next call reads it without a copy
};

OSG_D void store_through(double* p, double v) {   // agent scope: written through to memory, visible to every CU
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(v)),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
OSG_D double load_through(const double* p) {      // agent scope: never served from a stale L1 / L2 line
  return __longlong_as_double(static_cast<long long>(__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
}

constexpr int kSplitMaxA = 4;       // widest policy row the split kernel folds
constexpr int kSplitOwnerPath = 10;  // decision entries of a root path kept in registers
constexpr int kSplitRec = 1 + kSplitMaxA;  // doubles per member and buffer: own reach, regret terms
constexpr int kSplitChunk = 6;       // members whose records are requested together

struct GridCfr {
  Tree t;
  const int32_t* path_off;
  const int32_t* path;
  const int32_t* meta;         // [H] kind | nchild << 2 | (actor + 1) << 10
  const int32_t* info_player;  // [I]
  double* value;               // [H, P]
  double* dreg;                // [M, A]
  double* dpol;                // [M, A]
  int32_t* skip;               // [M]
  Tables tb;
  int M;
  const double* pol = nullptr; // [I, A] the policy a pass plays: tb.cur, or CFR-BR's effective policy (k_gcfr_effpol)
};

struct SubTree {
  int G, L, NL;                  // subtrees, cut level, padded histories per subtree
  const int32_t* nloc;           // [G] histories of the subtree
  const int32_t* desc;           // [G, NL] kind | nchild << 2 | level << 10 | (actor + 1) << 16
  const int32_t* fc;             // [G, NL] LOCAL index of the first child
  const int32_t* aux;            // [G, NL] decision: its index d among the subtree's decision histories; chance /
                                 //          terminal: the history's global index
  int ND;                        // padded decision histories per subtree
  const int32_t* ndec;           // [G]
  const int32_t* dec_row;        // [G, ND] info * A of decision history d
  const int32_t* mem_off;        // [G * P + 1] the subtree's members of player q: sub_mem[mem_off[g * P + q] ...)
  const int32_t* sub_rec;        // [., 8 + PL / 2 rounded up to 4] (the codes are 16-bit halves, 0xFFFF padded) per member: m (position in Tree::mem), its history's local index, its decision
                                 //   index | actions << 24, its first child's local index; the product of the chance
                                 //   probabilities on its root path (a double, path order), two unused words; then the decision
                                 //   entries of the path GROUPED BY PLAYER, PL / P codes per player in path order, -1 padded:
                                 //   (the ancestor's decision index in this subtree) * A + action index, i.e. an index into the
                                 //   policy rows the sweep has staged in LDS
  int PL;                        // codes per member: P groups of a multiple of 4, at most 4 kSubCodeChunks
  const int32_t* info_off;       // [P + 1] infostates of player q: info_list[info_off[q] ...)
  const int32_t* info_list;
  double* recbuf;                // [M, 8] the members' 64-byte records (kSubRecDoubles)
  int tree_barrier;              // 1: two-level arrival + release word; 0: round 4's flat counter
  unsigned int* bar;             // [0] arrival counter, [1] error flag, [2] release word, [16 + 16 g] group counters (zeroed per launch)
  unsigned int* host_err;        // pinned host word raised on a timeout: the next call reads it without a copy
  unsigned long long timeout_ticks;
  unsigned long long* stamps;    // null, or [P][5] wall-clock stamps of workgroup 0 in the launch's last iteration
  int stamp_wg = 0;              // the workgroup that writes the stamps (OSG_CFR_SUB_STAMPS = its index + 1)
  // Forest form (round 5): when there are more deal subtrees than resident workgroups (3-player leduc: 336 on 256), the
  // tree is cut ONE LEVEL DEEPER into pieces (the children of the deal roots) and the pieces are packed into one bin
  // per workgroup, balanced by size: "subtree g" above is then a forest of pieces in level order, every workgroup
  // sweeps ONE forest per pass and none takes two while the others wait.  The deal roots ("upper" histories) belong to
  // no forest: their policy rows ride in the forests' LDS rows (path codes), the pieces' root values leave through
  // root_value, and an upper member's terms are formed by the fold from those values (skip word = 2 + its index).
  const int32_t* nroot = nullptr;     // [G] piece roots of the forest; null: the bins are whole subtrees
  const int32_t* root_loc = nullptr;  // [G, NR] local index of piece root r
  const int32_t* root_idx = nullptr;  // [G, NR] its slot in root_value
  int NR = 0;
  double* root_value = nullptr;       // [histories of the pieces' level] the updating player's value of every piece root
  const int32_t* upper_rec = nullptr; // [U, 8] first child's slot in root_value, info * A, actions, 0, chance product (lo, hi), 0, 0
  // Round 5: what a pass does not have to fetch again.  The decision rows of a bin are ordered by acting player
  // (dec_off), so with one bin per workgroup (keep_rows) the policy rows STAY in LDS between passes and a pass fetches
  // only the rows the previous pass's fold rewrote (the previous updating player's) and the upper parents' rows; the
  // outcome probabilities of the bin's chance histories sit in LDS too (chance_prob: no trip to memory inside the level
  // loop); the fold takes its infostates from a packed descriptor (fold_info) in shares balanced by members (fold_off).
  const int32_t* dec_off = nullptr;   // [G, P + 2] rows of player q: [dec_off[q], dec_off[q + 1]); upper parents' rows from dec_off[P] to dec_off[P + 1]
  const double* chance_prob = nullptr;// [G, NCP] outcome probabilities of the bin's chance histories (aux = offset of the first)
  int NCP = 0;                        // (even)
  int keep_rows = 0;
  int lds_doubles = 0;                // dynamic LDS of the launch, in doubles: [policy rows ND * A | chance NCP | values NL | spare]
  const int32_t* fold_info = nullptr; // [infostates in info_list order, 4] infostate, actions, first member, members
  const double* term_val = nullptr;   // [G, P, NL] player q's return at every terminal history of the bin, local order (0 elsewhere)
  int prefetch = 1;                   // 0: nothing is fetched in the barriers' windows (measurement)
  const int32_t* fold_off = nullptr;  // [P, grid + 1] the share of workgroup w in pass q: entries [fold_off[q][w], fold_off[q][w + 1])
  // CFR-BR pass sets (k_cfr_sub<., kBr>, round 6): every infostate's best-response action index and acting player
  const int32_t* br_best = nullptr;   // [I] (null: plain CFR)
  const int32_t* br_player = nullptr; // [I]
};
OSG_D double readlane_f64(double v, int lane) {   // lane is wave-uniform: two v_readlane_b32, no LDS permute
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane(static_cast<int>(b), lane), hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), lane);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
OSG_D void store_through_i32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
OSG_D int32_t load_through_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte written-through stores / bypassing loads (buffer instructions with the sc1 bit: what the 8-byte agent-scope
// atomics above compile to, four words at a time; the compiler keeps the wait counters)
typedef unsigned int osg_u4 __attribute__((ext_vector_type(4)));
typedef double osg_d2 __attribute__((ext_vector_type(2)));
constexpr int kCachePolicySc1 = 16;
OSG_D __amdgpu_buffer_rsrc_t through_buffer(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
OSG_D void store_through16(__amdgpu_buffer_rsrc_t r, unsigned int byte_off, osg_d2 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(osg_u4, v), r, static_cast<int>(byte_off), 0, kCachePolicySc1);
}
OSG_D osg_u4 load_through16(__amdgpu_buffer_rsrc_t r, unsigned int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(byte_off), 0, kCachePolicySc1);
}
// The member records of k_cfr_sub (round 5): 64 bytes per member — regret terms [4], average-policy terms [4] — written
// by the member's thread as whole 16-byte pieces and fetched by the fold the same way (7 eight-byte stores and loads per
// member before).  A record that carries no terms says so in its first word: a quiet NaN whose low word is 1 (the member
// was pruned) or 2 + u (upper member u of the forest form: written once by the host, its terms are formed in the fold).
constexpr unsigned int kSubFlagHi = 0x7FF80000u;
constexpr int kSubRecDoubles = 8;
constexpr int kSubBarWords = 16 + 16 * 64;   // grid barrier words: a cooperative grid of up to 1 024 workgroups

constexpr int kSubThreads = 1024;
constexpr int kSubKD = 4;            // decision histories per thread: ND <= 4096
constexpr int kSubFoldInfos = 64;    // infostates a workgroup folds per round (one wavefront adds them up)
constexpr int kSubFoldX = 2;         // member records a thread fetches per round: <= 2048 per round
constexpr int kSubCodeChunks = 8;    // int4 chunks of path codes a member record holds at most (requested together)

struct EvalArrays {
  const int32_t* path_off;   // [M+1]
  const int32_t* path;
  const int32_t* info_level; // [I] tree level of the infostate's member histories
  const int32_t* mem_index;  // [H] member position m of a decision history, else -1
  int M;
  double* value;             // [H, P] scratch
  double* brv;               // [H] scratch
  double* cf;                // [M] scratch
  int32_t* best;             // [I] scratch: chosen action index
  double* out;               // [2P]: ev[P] then br[P]
  double* keep = nullptr;    // [H] or null: the best-response value of EVERY history for responder keep_r
  int keep_r = -1;           //          (TabularBestResponse::Value(history), best_response.h:127-128)
};

struct EvalJobs {
  int J, L, G, NT;            // jobs, cut level, subtrees (= histories on level L), histories on levels 0..L
  const int32_t* job;         // [J, 8] kind (0 expected returns, 1 + r best response of r), node_off, nodes, info_off,
                              //        infos, mem_off, members, -
  const int32_t* level_off;   // [J, D + 1] the job's histories of a level: a range of job-local indices
  const int32_t* node_desc;   // per job history: kind | nchild << 2 | (actor + 1) << 10
  const int32_t* node_fc;     //   job-local index of its first child
  const int32_t* node_row;    //   info * A of a decision node
  const int32_t* node_glob;   //   its index in the whole tree
  const int32_t* info_ent;    // per job infostate [4]: id, level, offset of its first member in the job's member list, members
  const int32_t* mem_ent;     // per job member [2]: member index m (position in Tree::mem), job-local history
  double* deal;               // expected returns [G, P], then best-response values [P, G]
  unsigned int* ticket;       // zero between launches
};

OSG_D void add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }  // hardware fp64 atomic (LDS and L2)
// The counter stream of an external-sampling trajectory by where the walk is (level = traverser nodes above, 0 .. 2).
OSG_HD uint64_t es_stream(int level, int b1, int b2) {
  return level == 0 ? 0u : (level == 1 ? 1u + static_cast<uint64_t>(b1) : 16u + 8u * static_cast<uint64_t>(b1) + static_cast<uint64_t>(b2));
}

#ifndef OSG_MCCFR_TREE_GLOBAL_DEFAULT
#define OSG_MCCFR_TREE_GLOBAL_DEFAULT 1   // round 6: 3.04e9 -> 4.23e9 trajectories/s at 16 x 2^20 (profiles/r06c_mccfr_tree_in_l2_ab.txt)
#endif
struct ResidentTree {
  const uint2* rec;      // [H]
  const double* uret;    // [K, P] distinct Returns() vectors
  const double* uprob;   // [nprob] distinct chance probabilities
  int K, nprob;
  int tree_global;       // 1: the traversals read the records from `rec` itself (read-only, L2-resident) and LDS holds
                         // the tables only, so that two workgroups fit a CU (leduc: 67 KB instead of 143 KB); 0: staged in LDS
};

constexpr int kMaxOsDepth = 32;

template <class T> struct TypeTag { using type = T; };

template <class T>
int upload(const std::vector<T>& v, T** d, hipStream_t stream) {
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  OSG_HIP(hipMalloc(reinterpret_cast<void**>(d), bytes));
  if (!v.empty()) {
    OSG_HIP(hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    // The callers pass vectors that die when they return, and a large copy from pageable memory may still be reading
    // the host buffer after the call (seen once as a GPU fault at a host address with an 87 MB vector): wait.
    if (v.size() * sizeof(T) > (64u << 10)) OSG_HIP(hipStreamSynchronize(stream));
  }
  return OSG_OK;
}

}  // namespace osg_cfr_impl

using namespace osg_cfr_impl;

struct osg_cfr {
  osg_ctx* ctx = nullptr;
  GameSpec spec;
  osg_cfr_cfg cfg{};
  int P = 0, H = 0, I = 0, A = 0, D = 0;
  int64_t n_chance = 0, n_decision = 0, n_terminal = 0;
  int max_level_width = 0;
  int average_type = 0;  // ES-MCCFR AverageType: 0 kSimple, 1 kFull (external_sampling_mccfr.h:48)
  int iteration = 0;
  const char* last_kernel = "";  // the kernel family the last iterate / sample call launched (osg_cfr_last_kernel)
  // host tree
  std::vector<int32_t> level_off, parent, first_child, info, mem_off, mem, nact, legal;
  std::vector<uint8_t> kind, nchild, aidx;
  std::vector<int32_t> edge_action;  // [H] the action (or chance outcome) on the edge from the parent, -1 at the root
  std::vector<int8_t> actor, info_player;
  std::vector<double> edge_prob, term_ret;
  std::vector<std::string> keys;
  // device tree
  int32_t *d_level_off = nullptr, *d_parent = nullptr, *d_first_child = nullptr, *d_info = nullptr,
          *d_mem_off = nullptr, *d_mem = nullptr, *d_nact = nullptr;
  uint8_t *d_kind = nullptr, *d_nchild = nullptr, *d_aidx = nullptr;
  int8_t *d_actor = nullptr, *d_info_player = nullptr;
  double *d_edge_prob = nullptr, *d_term_ret = nullptr;
  // device tables and work arrays
  double* d_tables = nullptr;  // regrets | cum | cur | dreg | dpol, each [I, A]
  double* d_reach = nullptr;   // [H, P+1]
  double* d_value = nullptr;   // [H, P]
  bool lds_resident = false;
  size_t lds_bytes = 0;
  // small-tree kernel: root paths of the decision histories (member order)
  std::vector<int32_t> path_off, path;
  int32_t *d_path_off = nullptr, *d_path = nullptr;
  bool small_tree = false;   // the all-in-LDS variant fits
  bool path_kernel = false;  // the path-based kernel (k_cfr_small) is usable at all
  int max_path_decisions = 0;  // the most decision entries any member's root path holds
  int first_decision_level = 0;  // the first level with a decision history (SmallTree::L0)
  size_t small_lds_bytes = 0;
  std::vector<int32_t> meta32, info_player32;
  int32_t *d_meta32 = nullptr, *d_info_player32 = nullptr, *d_skip = nullptr;
  double* d_node_delta = nullptr;  // dreg [M, A] | dpol [M, A]
  double* d_spare_delta[2] = {nullptr, nullptr};  // osg_mccfr_spare_delta_buffer: [2, I, A] each, allocated on request
  bool delta_clean[3] = {false, false, false};    // the internal / spare delta buffers are all zero (the last fold left them so)
  // one workgroup per deal subtree (k_cfr_split)
  bool split_ok = false, split_br_ok = false;
  int split_G = 0, split_L = 0, split_NL = 0, split_NM = 0, split_NI = 0, split_threads = 0;
  size_t split_lds_bytes = 0;
  int32_t *d_split_nloc = nullptr, *d_split_desc = nullptr, *d_split_fc = nullptr, *d_split_row = nullptr,
          *d_split_glob = nullptr, *d_split_mem_m = nullptr, *d_split_mem_hloc = nullptr, *d_split_info = nullptr;
  double* d_split_terms = nullptr;
  unsigned int* d_split_bar = nullptr;
  // one cooperative launch, a workgroup per deal subtree of any size (k_cfr_sub)
  bool sub_ok = false;
  bool sub_br_ok = false;           // k_cfr_sub<., kBr> (the CFR-BR pass set) fits the same grid
  int sub_G = 0, sub_L = 0, sub_NL = 0, sub_K = 0, sub_grid = 0;
  size_t sub_lds_bytes = 0;
  int sub_ND = 0, sub_PL = 0;
  int32_t *d_sub_ndec = nullptr, *d_sub_dec_row = nullptr, *d_sub_rec = nullptr;
  int32_t *d_sub_nloc = nullptr, *d_sub_desc = nullptr, *d_sub_fc = nullptr, *d_sub_aux = nullptr, *d_sub_mem_off = nullptr,
          *d_sub_info_off = nullptr, *d_sub_info_list = nullptr;
  unsigned int* d_sub_bar = nullptr;
  // forest form of k_cfr_sub (SubTree's comment): the kernel's own skip words, piece roots, upper members
  bool sub_forest = false;
  int sub_NR = 0, sub_G0 = 0;   // G0: the deal subtrees; sub_G: the bins they (or their pieces) were packed into
  unsigned long long *d_sub_stamps = nullptr, *d_mccfr_stamps = nullptr;   // profiling stamps (per solver: never shared across contexts / devices)
  double *d_sub_recbuf = nullptr, *d_sub_chance_prob = nullptr, *d_sub_term_val = nullptr;
  int32_t *d_sub_dec_off = nullptr, *d_sub_fold_info = nullptr, *d_sub_fold_off = nullptr;
  int sub_NCP = 0;
  bool sub_keep_rows = false;
  int32_t *d_sub_nroot = nullptr, *d_sub_root_loc = nullptr, *d_sub_root_idx = nullptr, *d_sub_upper_rec = nullptr;
  double* d_sub_root_value = nullptr;
  unsigned int* h_sub_err = nullptr;   // pinned: raised by the kernel when a grid barrier times out
  // policy evaluation (k_policy_eval)
  std::vector<int32_t> info_level, mem_index;
  bool eval_ok = true;  // every infostate's members sit on one tree level
  int32_t *d_info_level = nullptr, *d_mem_index = nullptr, *d_best = nullptr;
  double *d_eval = nullptr;  // value [H,P] | brv [H] | cf [M] | out [2P] | policy [I,A]
  double* d_eval_ev = nullptr;   // [H, P]: the expected returns of the large-tree evaluation (allocated on first use)
  std::vector<int32_t> eval_level_off;   // [D + 1] the infostates of level l: d_eval_level_info[eval_level_off[l] ...)
  int32_t* d_eval_level_info = nullptr;
  int32_t* d_eval_level_off = nullptr;   // the same offsets on the device (k_geval_persist)
  unsigned int* d_geval_bar = nullptr;   // its grid barrier's counters
  int geval_grid = 0;                    // its resident grid (-1: none, a launch per level and phase instead)
  const char* last_eval_kernel = "";
  // the evaluation as independent jobs over the device (k_eval_jobs)
  bool jobs_ok = false;
  int jobs_J = 0, jobs_L = 0, jobs_G = 0, jobs_NT = 0, jobs_threads = 0;
  size_t jobs_lds_bytes = 0;
  int32_t *d_jobs_job = nullptr, *d_jobs_level = nullptr, *d_jobs_desc = nullptr, *d_jobs_fc = nullptr, *d_jobs_row = nullptr,
          *d_jobs_glob = nullptr, *d_jobs_info = nullptr, *d_jobs_mem = nullptr;
  double* d_jobs_deal = nullptr;
  unsigned int* d_jobs_ticket = nullptr;
  double* h_eval_out = nullptr;  // pinned, mapped: the evaluation kernels write their [2 P] results here
  // LDS-resident MCCFR traversal (k_mccfr_resident)
  bool resident_ok = false;
  size_t resident_lds_bytes = 0;
  int n_uret = 0, n_uprob = 0, num_cus = 0;
  uint64_t* d_rec = nullptr;
  double *d_uret = nullptr, *d_uprob = nullptr;

  Tree tree() const {
    Tree t;
    t.H = H; t.I = I; t.A = A; t.P = P; t.D = D;
    t.level_off = d_level_off; t.parent = d_parent; t.first_child = d_first_child; t.kind = d_kind;
    t.nchild = d_nchild; t.aidx = d_aidx; t.actor = d_actor; t.info = d_info; t.edge_prob = d_edge_prob;
    t.term_ret = d_term_ret; t.mem_off = d_mem_off; t.mem = d_mem; t.nact = d_nact; t.info_player = d_info_player;
    return t;
  }
  // Replicas: B independent solvers of the same tree (tables [B][5][I, A]); `selected` is the one the
  // table accessors / evaluation look at.
  int B = 1, selected = 0;
  size_t replica_stride() const { return 5 * static_cast<size_t>(I) * A; }
  double* replica_base(int r) const { return d_tables + static_cast<size_t>(r) * replica_stride(); }
  double* regrets() const { return replica_base(selected); }
  double* cum() const { return replica_base(selected) + static_cast<size_t>(I) * A; }
  double* cur() const { return replica_base(selected) + 2 * static_cast<size_t>(I) * A; }
  double* dreg() const { return replica_base(selected) + 3 * static_cast<size_t>(I) * A; }
  double* dpol() const { return replica_base(selected) + 4 * static_cast<size_t>(I) * A; }
};

namespace osg_cfr_impl {
// ---- osg_cfr.hip ----
int cfr_sub_error(const osg_cfr* s);   // a grid barrier timed out in an earlier launch: the solver refuses further work
// ---- osg_cfr_small.hip ----
void cfr_small_prepare(osg_cfr* s);    // LDS caps of k_cfr<true> / k_cfr_small (clears lds_resident / small_tree where refused)
int cfr_small_iterate(osg_cfr* s, Tables tb, int iters, int threads, unsigned grid_b);   // k_cfr_small / k_cfr
void cfr_general_br_pass(osg_cfr* s, Tables tb, int threads, osg_cfr_cfg cfg);           // k_cfr<false, kBr> x 1
// ---- osg_cfr_split.hip ----
int build_split(osg_cfr* s);
int launch_split(osg_cfr* s, SmallTree stree, SplitTree sp, Tables tb, int iters, int iteration0, osg_cfr_cfg cfg, bool br);
// ---- osg_cfr_sub.hip ----
int build_sub(osg_cfr* s);
int cfr_sub_iterate(osg_cfr* s, Tables tb, int iters);      // k_cfr_sub
int cfr_grid_iterate(osg_cfr* s, Tables tb, int iters);     // k_gcfr_*
int cfr_grid_br_iterate(osg_cfr* s, Tables tb, const EvalArrays& ea, osg_cfr_cfg cfg, int iters);   // CFR-BR on large trees, a launch per phase
int cfr_sub_br_iterate(osg_cfr* s, Tables tb, const EvalArrays& ea, osg_cfr_cfg cfg, int iters);    // CFR-BR on large trees, k_cfr_sub<., kBr>
// ---- osg_cfr_eval.hip ----
int build_eval_jobs(osg_cfr* s);
EvalJobs eval_jobs_of(const osg_cfr* s);
bool eval_takes_the_grid(const osg_cfr* s);
bool OSG_EVAL_JOBS_ENABLED();
int launch_grid_eval(const osg_cfr* s, const EvalArrays& ea, const double* src, bool from_cum, double* d_pol, bool only_br);
int cfr_best_responses_to_current(osg_cfr* s, const EvalArrays& ea, int threads, bool jobs);   // every player's best response (CFR-BR)
// ---- osg_cfr_mccfr.hip ----
int build_resident_tree(osg_cfr* s);
}  // namespace osg_cfr_impl
