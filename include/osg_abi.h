/* =============================================================================
 * osg_abi.h — C-ABI of the MI355X batched game-step and search engine.
 *
 * This is the drop-in boundary for OpenSpiel's data-parallel hot path: every
 * entry point below is what a binding of the reference (pybind / cgo / Rust
 * FFI) would call instead of looping over per-state virtual calls.  Plain
 * pointers and sizes only — no torch, no C++ types.  Conventions follow the
 * reference's in-tree C API (open_spiel/rust/src/rust_open_spiel.h:18-88:
 * opaque handles, caller-supplied out-buffers).
 *
 * All functions return 0 on success and a negative osg_status on failure; the
 * message is available from osg_last_error() (thread-local).  Nothing in the
 * library calls exit() or throws across the boundary; the C++ host classes
 * (open_spiel_amd/csrc/host) turn a non-zero code into SpielFatalError so the
 * observable error behaviour matches open_spiel/spiel_utils.cc:119-137.
 *
 * Pointers named d_* are DEVICE pointers (HBM, e.g. torch tensor.data_ptr());
 * pointers named h_* are host pointers.  Buffers named without prefix take an
 * `on_host` flag.  The library owns the memory behind handles; the caller owns
 * every buffer it passes in.  An osg_ctx is bound to one device and one HIP
 * stream and is not thread-safe; distinct contexts are independent.  All work
 * is enqueued on the context's stream; functions that write to host buffers
 * synchronise that stream before returning, device-only functions do not.
 * =========================================================================== */
#ifndef OSG_ABI_H_
#define OSG_ABI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  OSG_OK = 0,
  OSG_ERR_INVALID = -1,     /* bad argument / malformed game string           */
  OSG_ERR_UNSUPPORTED = -2, /* game or parameter not available on the device   */
  OSG_ERR_HIP = -3,         /* a HIP runtime call failed                       */
  OSG_ERR_ILLEGAL = -4,     /* an illegal action was applied (count > 0)       */
  OSG_ERR_NOMEM = -5
} osg_status;

typedef struct osg_ctx osg_ctx;     /* one per (process, device): HIP stream      */
typedef struct osg_batch osg_batch; /* N states of one game, SoA in HBM           */
typedef struct osg_cfr osg_cfr;     /* a flattened game tree + fp64 CFR tables    */

/* Sentinels (open_spiel/spiel_globals.h:26-56,82). */
#define OSG_CHANCE_PLAYER (-1)
#define OSG_TERMINAL_PLAYER (-4)
#define OSG_INVALID_ACTION (-1)

/* Replaces Game::NumDistinctActions / MaxChanceOutcomes / NumPlayers /
 * ObservationTensorShape / InformationStateTensorShape / MaxGameLength /
 * MinUtility / MaxUtility (open_spiel/spiel.h:927-1255). */
typedef struct {
  int32_t game_kind;           /* 0 ttt, 1 connect_four, 2 hex, 3 kuhn, 4 leduc */
  int32_t num_players;
  int32_t num_distinct_actions;
  int32_t max_chance_outcomes;
  int32_t max_game_length;
  int32_t max_chance_nodes;
  int32_t obs_size;            /* ObservationTensorSize()                        */
  int32_t info_size;           /* InformationStateTensorSize(), 0 if absent      */
  int32_t obs_shape[4];        /* rank in obs_rank                               */
  int32_t obs_rank;
  int32_t info_shape[4];
  int32_t info_rank;
  int32_t mask_words;          /* u32 words per legal mask = ceil(max(A,C)/32)   */
  int32_t compact_mask_bytes;  /* bytes per state of the fused step's mask: 1,2,4*mask_words */
  int32_t state_words;         /* SoA planes per state                           */
  int32_t state_word_bytes;    /* 4 or 8: bytes per plane element                */
  double min_utility;
  double max_utility;
  char canonical[128];         /* Game::ToString(): the string as parsed         */
} osg_game_desc;

const char* osg_last_error(void);

/* ---- context ------------------------------------------------------------ */
/* own_stream != 0: the library creates (and owns) a non-blocking stream and
 * `stream` is ignored.  own_stream == 0: `stream` is a hipStream_t owned by the
 * caller (e.g. torch's current stream; NULL is the device's default stream).
 * The engine is written for one process per GPU: osg_ctx_create makes `device` the calling
 * thread's current device and later calls launch on the context's stream without selecting it
 * again — a process that holds contexts on several devices makes the context's device current
 * (hipSetDevice) on the calling thread before each call. */
int osg_ctx_create(int device, void* stream, int own_stream, osg_ctx** out);
int osg_ctx_destroy(osg_ctx* ctx);
int osg_ctx_synchronize(osg_ctx* ctx);
void* osg_ctx_stream(osg_ctx* ctx);
/* Give back what the context caches between calls: the MCTS node pool (grow-only otherwise: a search without a
 * node budget sizes it at 1 + max_simulations x widest-node slots per root, up to 60 % of the free HBM) and the
 * staging buffer.  Waits for the stream first.  The next search allocates again. */
int osg_ctx_trim(osg_ctx* ctx);
/* Rebind a context created on a caller's stream (own_stream == 0) to another stream of the same device, e.g. the
 * stream a hipGraph is being captured on: every later call of the context's objects is issued there. */
int osg_ctx_set_stream(osg_ctx* ctx, void* stream);

/* ---- game description (no device needed) -------------------------------- */
/* Replaces LoadGame(game_string) (open_spiel/spiel.cc:255) for the five hot-path
 * games; unknown parameters / wrong types are OSG_ERR_INVALID as in
 * spiel.cc:65-90, games or sizes without a device layout OSG_ERR_UNSUPPORTED. */
int osg_game_describe(const char* game_string, osg_game_desc* out);

/* ---- batches of states --------------------------------------------------- */
/* Replaces Game::NewInitialState() x n (spiel.h:943-950). */
int osg_batch_create(osg_ctx* ctx, const char* game_string, int64_t n, osg_batch** out);
int osg_batch_destroy(osg_batch* b);
int64_t osg_batch_size(const osg_batch* b);
int osg_batch_describe(const osg_batch* b, osg_game_desc* out);
int osg_batch_reset(osg_batch* b);                              /* all -> initial state  */
/* State::Clone() for a whole batch (spiel.h:737). Same game and size. */
int osg_batch_copy(osg_batch* dst, const osg_batch* src);
/* dst[i] = src[index[i]] (dst size = number of indices): Clone() of chosen states. */
int osg_batch_gather(osg_batch* dst, const osg_batch* src, const int64_t* index, int on_host);
/* Raw SoA image: state_words planes of n elements each (plane-major). */
int osg_batch_download(const osg_batch* b, void* h_words);
int osg_batch_upload(osg_batch* b, const void* h_words);
/* State `index` of the batch := the position with these cells, one character per cell ('.', 'x', 'o').  tic_tac_toe: 9
 * cells, cell a = action a (TicTacToeState(game, TicTacToeStateStruct), games/tic_tac_toe/tic_tac_toe.cc:273-336);
 * connect_four: rows x cols cells, cell r * cols + c with row 0 the BOTTOM row (ConnectFourStateStruct::board;
 * ConnectFourState(game, struct, strict_validation) and ConnectFourState(game, string),
 * games/connect_four/connect_four.cc:352-470).  The position is built on the device with the game's own rules
 * (outcome recomputed; player to move = parity of the stone count, which is how this layout stores it — the
 * piece-count rules of the reference's constructors are the caller's to enforce).  OSG_ERR_INVALID: wrong cell count,
 * another character, a gap in a connect_four column, both players with a line; OSG_ERR_UNSUPPORTED: other games. */
int osg_batch_set_cells(osg_batch* b, int64_t index, const char* cells, int n_cells);
/* Device pointer to the SoA planes (plane k at base + k * n elements). */
void* osg_batch_device_ptr(osg_batch* b);

/* State::LegalActionsMask() (spiel.cc:518-524), bit-packed: mask[i*W + a/32] bit a%32;
 * at chance nodes the bits are the legal chance outcomes; all zero when terminal. */
int osg_legal_mask(const osg_batch* b, uint32_t* mask, int on_host);

/* State::ApplyAction (spiel.cc:441-451 + each game's DoApplyAction), in place.
 * actions[i] == -1 leaves state i untouched.  Illegal actions (not in
 * LegalActions(), or any action on a terminal state) leave the state untouched
 * and are counted; the count is returned through *illegal (may be NULL; then a
 * non-zero count turns into OSG_ERR_ILLEGAL at the next synchronising call). */
int osg_apply(osg_batch* b, const int32_t* actions, int on_host, int64_t* h_illegal);

/* IsTerminal / CurrentPlayer / Returns (per game; e.g. connect_four.cc:122-128,
 * 277-285).  Any pointer may be NULL.  returns is [n, num_players] fp64. */
int osg_status_query(const osg_batch* b, int8_t* cur_player, uint8_t* terminal,
                     double* returns, int on_host);

/* State::ChanceOutcomes() probabilities for every outcome id: probs[n, max_chance]
 * fp64, 0 for outcomes that are not legal (or at non-chance nodes). */
int osg_chance_probs(const osg_batch* b, double* probs, int on_host);

/* The fused headline kernel: LegalActions check + ApplyAction + IsTerminal /
 * CurrentPlayer / outcome + LegalActions of the successor, one pass over the
 * SoA state.  Device pointers only; src may equal dst.
 *   d_actions [n] u8   action id (0xFF = skip)
 *   d_mask    [n * compact_mask_bytes] legal mask of the successor state; NULL = do not write it: hex boards of up
 *                      to 128 cells only (hex.cc:280-293: the successor's legal actions are its empty cells, i.e.
 *                      ~occupied of the record the step writes), OSG_ERR_UNSUPPORTED for every other game.
 *                      A caller deriving that mask from the raw words must (i) AND it with the board's cell bits — in
 *                      the folded record (hex 9x9, 11x11, 19x19: osg_game_desc.state_words == 4 planes x words, no meta
 *                      word) the top 5 bits of every plane's LAST word (bits 27-31) carry mover / result / ply / first
 *                      move, not cells: mask that word with 0x07FFFFFF — and (ii) treat a terminal state (status bit7)
 *                      as having no legal action.  osg_legal_mask does both.
 *   d_status  [n] u8   bit7 terminal | bit6 action was illegal (state unchanged) |
 *                      not terminal: bits0-3 = current player + 1 (0 = chance) |
 *                      terminal:     bits0-2 = outcome (board games: 0 p0 wins,
 *                      1 p1 wins, 2 draw; poker: 7 = see osg_status_query) */
int osg_step(const osg_batch* src, osg_batch* dst, const uint8_t* d_actions,
             void* d_mask, uint8_t* d_status);

/* State::ObservationTensor(player) / InformationStateTensor(player)
 * (spiel.cc:908-945 + each game; observer.h:174-185 zero-fill semantics).
 * which: 0 observation, 1 information state.  player in [0, P), or -1 = the
 * state's current player (player 0 where that is chance/terminal).
 * out is [n, size] fp32. */
int osg_observation(const osg_batch* b, int player, int which, float* out, int on_host);

/* State::InformationStateString(player) of state `index` (kuhn_poker.cc:109-166,
 * leduc_poker.cc:198-239) — the key of the CFR tables.  Host formatter over the packed
 * state words; returns the length (excluding NUL) or <0. */
int osg_information_state_string(const osg_batch* b, int64_t index, int player, char* buf, int cap);

/* State::ObservationString(player) of state `index`: the board (tic_tac_toe.cc:163-175,
 * connect_four.cc:212-222, hex.cc:341-359) or the imperfect-recall observer string of the poker games
 * (kuhn_poker.cc:109-166, leduc_poker.cc:198-239).  Host formatter over the packed state words; returns
 * the length (excluding NUL) or <0. */
int osg_observation_string(const osg_batch* b, int64_t index, int player, char* buf, int cap);

/* State::ToString() of state `index` (tic_tac_toe.cc:163-175, connect_four.cc:212-222, hex.cc:341-359,
 * kuhn_poker.cc:253-268, leduc_poker.cc:463-496) and State::ActionToString(player, action)
 * (tic_tac_toe.cc:266-270, connect_four.cc:158-161, hex.cc:295-314, kuhn_poker.cc:244-251,
 * leduc_poker.cc:459-461); player -1 = chance.  Host formatters; return the length or <0. */
int osg_state_string(const osg_batch* b, int64_t index, char* buf, int cap);
int osg_action_string(const osg_batch* b, int64_t index, int player, int32_t action, char* buf, int cap);

/* Plain device-to-device copy of `bytes` (a multiple of 16; both pointers 16-byte aligned) on the
 * context's stream with 16-byte accesses per lane and non-temporal stores (the fastest plain copy measured
 * here: 91.6 us for 587 MB against 96.1 us with ordinary stores): the memory-only ceiling the step / tensor kernels
 * are measured against (bench.py `roofline.copy_ceiling`; SURVEY.md 8(d) "measured device-copy
 * bandwidth as secondary denominator").  No reference counterpart. */
int osg_copy_bytes(osg_ctx* ctx, void* d_dst, const void* d_src, int64_t bytes);

/* Environment loop on device: `steps` times { sample a uniformly random legal
 * action (chance outcomes by their distribution), apply, auto-reset terminal
 * states to the initial state }.  d_counters[0] += env steps applied,
 * d_counters[1] += episodes finished.  RNG stream = (seed, global index). */
int osg_random_steps(osg_batch* b, uint64_t seed, int64_t index_offset, int steps,
                     unsigned long long* d_counters);

/* SURVEY.md 8(d) synthetic benchmark inputs on the counter stream (seed, index_offset + i): state i = the
 * initial state advanced by depth_i = draw mod depth_mod uniformly random legal moves (chance outcomes by
 * their distribution), the trajectory re-drawn from the same stream if it ends earlier, so every state is
 * non-terminal; d_actions [n] u8 (may be NULL) = one more uniformly random legal action of that state,
 * d_depth [n] i32 (may be NULL) = depth_i.  The batch a benchmark times can therefore be regenerated by the
 * CPU oracle state for state (oracle/spiel_oracle_capi.cpp osgo_synth_batch restates the loop; the rules it
 * runs are the reference's: spiel.cc:441-451, connect_four.cc:130-156).  depth_mod in [1, MaxGameLength()];
 * a depth no trajectory survives (e.g. >= the shortest game) degrades to the initial state after 2^14 attempts.
 * No reference counterpart: the reference's benchmarks build their states one by one on the host. */
int osg_synth_batch(osg_batch* b, uint64_t seed, int64_t index_offset, int depth_mod, uint8_t* d_actions,
                    int32_t* d_depth);

/* One reinforcement-learning environment step for every state (python/rl_environment.py:379-418
 * Environment.step + get_time_step; replaces the Python loop of python/vector_env.py:51-54).
 * Device pointers only.  d_should_reset [n] u8 is in/out: an environment flagged 1 starts a new
 * episode and ignores its action; otherwise d_actions[i] is applied (-1: environment left as it is).
 * Chance nodes are then
 * resolved by sampling from the counter stream (seed, index_offset + i, step_index).
 * Outputs: d_cur_player [n] i8, d_step_type [n] u8 (0 FIRST, 1 MID, 2 LAST), d_rewards [n, P]
 * f64 (terminal returns at LAST, zeros otherwise), d_mask [n, mask_words] u32 legal mask of the
 * new state; d_should_reset becomes 1 exactly where the step type is LAST.  Illegal actions leave
 * the state unchanged and are reported by osg_ctx_synchronize (OSG_ERR_ILLEGAL). */
int osg_env_step(osg_batch* b, const int32_t* d_actions, uint8_t* d_should_reset, uint64_t seed, int64_t index_offset,
                 int64_t step_index, int8_t* d_cur_player, uint8_t* d_step_type, double* d_rewards, uint32_t* d_mask);
/* The same step with compact side arrays for hosts that keep their own TimeStep layout (no reference counterpart: the
 * reference's TimeStep carries int actions and float64 rewards, python/rl_environment.py:257-318 — 20 of the 60 bytes
 * osg_env_step moves per connect_four environment; this form moves 41).  d_actions [n] u8: the action, 0xFF = leave the
 * environment as it is.  d_flags [n] u8 is in/out: bits 0-1 the step type (0 FIRST, 1 MID, 2 LAST), bits 2-7 the current
 * player + 4 (chance -1 -> 3, terminal -4 -> 0); an environment whose flag byte says LAST on input starts a new episode
 * and ignores its action (initialise the array to 2 to start every environment).  d_rewards_x2 [n, P] i8 = TWICE the
 * reward (the returns of the games served are multiples of 0.5): terminal returns at LAST, zeros otherwise; a game whose
 * doubled returns do not fit a signed byte (leduc_poker with 6+ players), or with more than 255 actions (hex from 16 x 16),
 * answers OSG_ERR_UNSUPPORTED.  d_mask, the chance
 * sampling, the counter streams and the illegal-action report are osg_env_step's: the two forms step identically. */
int osg_env_step_compact(osg_batch* b, const uint8_t* d_actions, uint8_t* d_flags, uint64_t seed, int64_t index_offset,
                         int64_t step_index, int8_t* d_rewards_x2, uint32_t* d_mask);

/* algorithms::RandomRolloutEvaluator::Evaluate (open_spiel/algorithms/mcts.cc:43-72)
 * for every root: n_rollouts uniform-random playouts to the end of the game.
 * sum_returns [n, P] fp64 = SUM over rollouts of Returns() (divide by n_rollouts
 * for Evaluate's mean); steps[n] i32 (may be NULL) = plies played.  Rollout r of
 * root i draws from the counter stream (seed, index_offset + i, r).  hex with steps == NULL: the playouts place their
 * stones with the same draws until the board is full and read the winner off the filled board (a hex game cannot be
 * un-won) — the same sums, ~5 x the rate of the move-by-move rules, which a non-NULL steps keeps.
 * A playout is cut off after 512 moves (no game served here lasts longer than 130): a record the
 * rules cannot finish — uploaded, neither terminal nor with a legal action — contributes Returns()
 * of a running game (zeros) instead of spinning on the device.  hex boards with a single row or
 * column are refused (OSG_ERR_UNSUPPORTED): with the reference's `else if` between a colour's two
 * edges (hex.cc:122-126,146-150) one colour can never win there, so playouts would not end. */
int osg_rollout(const osg_batch* roots, uint64_t seed, int64_t index_offset, int n_rollouts,
                double* sum_returns, int32_t* steps, int on_host);

/* algorithms::MCTSBot (open_spiel/algorithms/mcts.{h,cc}) for every root of the
 * batch, one wavefront per root.  Fields as MCTSBot's constructor (mcts.h:161-169).
 * The same two guards as osg_rollout apply (playout length, one-row / one-column hex boards);
 * a node that is not terminal and has no legal action is evaluated as a leaf, never expanded. */
typedef struct {
  double uct_c;
  int32_t max_simulations;
  int32_t n_rollouts;       /* RandomRolloutEvaluator(n_rollouts, seed)            */
  int32_t solve;            /* MCTS-Solver backup (mcts.cc:398-434)                */
  int32_t max_nodes;        /* > 0: MCTSBot's max_nodes_ = (max_memory_mb << 20) / sizeof(SearchNode) + 1
                               (mcts.cc:214): when a tree reaches it, every node visited fewer than
                               gc_limit_ times loses its children (GarbageCollect, mcts.cc:441-482).
                               <= 0: no caller limit: the pool gets 1 + max_simulations x widest-node slots
                               per root (it can never run out) when that fits 60 % of the free HBM, else
                               what fits, collected the same way at that size.  The pool stays cached in
                               the context (grow-only) until osg_ctx_trim / osg_ctx_destroy          */
  uint64_t seed;
  int64_t index_offset;     /* global index of root 0 (multi-GPU sharding)         */
  int32_t layout;           /* 0 auto; 1 one LANE per root (64 searches per wavefront,
                               sequential rollouts); 2 one WAVEFRONT per root (children
                               and rollouts spread over the 64 lanes; hex playouts as a
                               wave-parallel random fill).  The two layouts define their
                               random streams differently (see osg_common.h), so results
                               are reproducible per layout, not across layouts.  Layout 2
                               serves games of up to 128 actions and (round 6) the hex boards
                               above 128 cells WITHOUT the swap rule (to 19 x 19: 3 / 4 / 6
                               64-cell sets per colour in scalar registers; 0 picks it there —
                               2.1-2.6 x layout 1 on 12 x 12 ... 16 x 16); hex above 128 cells
                               with the swap rule, connect_four above 64 board bits and
                               leduc_poker with 4+ players are searched with layout 1 (0 picks
                               it), and layout 2 answers OSG_ERR_UNSUPPORTED for them.  A node
                               holds up to 511 actions.  Layout 2 is TUNED for hex without the
                               swap rule (what 0 picks it for); its instantiations for the other
                               games are correct (replay parity in the tests) but spill 76-129
                               vector registers (profiles/r05_kernel_resources.txt): use layout 1
                               for tic_tac_toe, connect_four, kuhn_poker and leduc_poker      */
  int32_t child_selection_policy;  /* ChildSelectionPolicy (mcts.h:148): 0 UCT (mcts.cc:90-101),
                               1 PUCT (mcts.cc:103-112) with the evaluator's prior — uniform over
                               the legal actions for RandomRolloutEvaluator (mcts.cc:74-87)        */
} osg_mcts_cfg;
/* Outputs (host or device by on_host; any may be NULL):
 *   best_action [n] i32            SearchNode::BestChild().action (mcts.cc:127-143)
 *   child_visits [n, A] i32        explore_count of the root child for action a
 *   child_reward [n, A] f64        total_reward of that child
 *   child_outcome [n, A] i8        proven outcome for the root player (-1,0,1) or 2 = unproven,
 *                                  3 = no such child
 *   root_stats [n, 4] f64          root explore_count, nodes used, root outcome for
 *                                  the root player (NaN if unproven), simulations run */
int osg_mcts_search(const osg_batch* roots, const osg_mcts_cfg* cfg, int32_t* best_action,
                    int32_t* child_visits, double* child_reward, int8_t* child_outcome,
                    double* root_stats, int on_host);

/* ---- MCTS with the Evaluator outside the kernel (Evaluator interface mcts.h:83-92; the batched shape of
 * alpha_zero_torch/vpevaluator.{h,cc}) ------------------------------------------------------------------
 * Search trees for every root of a batch that persist between calls.  osg_mcts_tree_advance runs every
 * search until it needs its evaluator and reports, per root, in d_request [n] u8:
 *   0 finished (max_simulations run, root proven, or a single root child: mcts.cc:361-366,437-440)
 *   1 wants Prior(state) of the node it is about to expand (mcts.cc:281-283) — only with flag 1;
 *     5 (= 1 | 4) when that node is the search's root (where MCTSBot mixes in Dirichlet noise, mcts.cc:284-292)
 *   2 wants Evaluate(state) of the leaf it has reached (mcts.cc:377-380)
 *   3 paused by max_new_simulations (more simulations to run; nothing wanted)
 * The state a request refers to is written to element i of `leaf` (a batch of the roots' game and size, which
 * also keeps the parked searches' working states: do not modify it between calls).  The next call takes the
 * answers: d_prior [n, num_distinct_actions] f64 (probability of action a at [i, a]; read for roots that
 * reported 1) and d_value [n, num_players] f64 (read for roots that reported 2); either may be NULL when no
 * root reported that request (a root whose answer is missing stays parked and reports its request again).  h_counts (may be NULL) receives the number of roots per request code [4].
 * cfg as for osg_mcts_search (layout ignored: one lane per root; the tree-policy streams are those of layout 1,
 * so with osg_mcts_tree_rollout_values as the evaluator the search is osg_mcts_search's, draw for draw).
 * flags: 1 = priors come from the caller (else uniform over the legal actions, RandomRolloutEvaluator::Prior
 * mcts.cc:74-87; chance nodes always use their ChanceOutcomes()); 2 = dont_return_chance_node (mcts.h:168);
 * 4 = leaves are evaluated inside the launch by RandomRolloutEvaluator(cfg.n_rollouts, cfg.seed) on the streams of
 * osg_mcts_tree_rollout_values: no request 2 is ever reported, and without flag 1 a whole search is one call;
 * 8 (with 1) = every answer to a value request comes WITH the prior of the same state in d_prior (one network forward
 * gives both: alpha_zero_torch/vpevaluator.cc:60-85 caches them per state): the prior is kept until the leaf is
 * expanded on its second visit, so a simulation is ONE evaluator round instead of up to two; a node whose children a
 * garbage collection cleared asks for its prior again (request 1).  d_prior must then be non-NULL in every call. */
typedef struct osg_mcts_tree osg_mcts_tree;
int osg_mcts_tree_create(const osg_batch* roots, const osg_mcts_cfg* cfg, int flags, osg_mcts_tree** out);
int osg_mcts_tree_destroy(osg_mcts_tree* t);
int osg_mcts_tree_advance(osg_mcts_tree* t, osg_batch* leaf, const double* d_prior, const double* d_value,
                          uint8_t* d_request, int max_new_simulations, int64_t* h_counts);
/* The same for a host that holds no device memory: answers and requests in host arrays (staged through buffers
 * the tree owns); values_on_device != 0: the values were left in the tree's own buffer by
 * osg_mcts_tree_rollout_values(t, leaf, NULL). */
int osg_mcts_tree_advance_host(osg_mcts_tree* t, osg_batch* leaf, const double* h_prior, const double* h_value,
                               int values_on_device, uint8_t* h_request, int max_new_simulations, int64_t* h_counts);
/* RandomRolloutEvaluator::Evaluate (mcts.cc:43-72) of every leaf that reported request 2, on the streams
 * of osg_mcts_search: mean Returns() of cfg.n_rollouts playouts into d_value [n, num_players] (device; NULL = the
 * tree's own value buffer, see osg_mcts_tree_advance_host). */
int osg_mcts_tree_rollout_values(osg_mcts_tree* t, const osg_batch* leaf, double* d_value);
/* The outputs of osg_mcts_search (device pointers, any may be NULL) plus child_prior [n, A] f64. */
int osg_mcts_tree_results(osg_mcts_tree* t, int32_t* best_action, int32_t* child_visits, double* child_reward,
                          int8_t* child_outcome, double* child_prior, double* root_stats);
/* One root's whole tree for the host (SearchNode, mcts.h:114-146): osg_mcts_tree_nodes = nodes in use;
 * download fills host arrays of that length: meta (action [0:8) | player + 1 [8:12) | children [12:20) |
 * has outcome [20] | outcome of player 0 + 1 [21:23) | terminal [23]), index of the first child (children are
 * contiguous), explore_count, total_reward, prior. */
/* The actions from root `root` to the node its search is parked at (the state of its pending request), for a
 * host that wants the State with its history.  Returns the length (<= cap) or a negative status. */
int osg_mcts_tree_leaf_path(osg_mcts_tree* t, int64_t root, int32_t* h_actions, int cap);
int64_t osg_mcts_tree_nodes(osg_mcts_tree* t, int64_t root);
int osg_mcts_tree_download(osg_mcts_tree* t, int64_t root, int64_t cap, uint32_t* h_meta, uint32_t* h_first,
                           uint32_t* h_count, double* h_total, double* h_prior);

/* ---- tabular CFR family -------------------------------------------------- */
typedef struct {
  int32_t alternating_updates;   /* CFRSolverBase ctor (cfr.h:190-196)            */
  int32_t linear_averaging;
  int32_t regret_matching_plus;
  int32_t solver;                /* 0: CFRSolverBase family, tables start at 0 (cfr.h:47-52);
                                    1: ExternalSamplingMCCFRSolver, regrets and cumulative policy
                                       start at kInitialTableValues = 1e-6
                                       (external_sampling_mccfr.h:59, .cc:142-143);
                                    2: OutcomeSamplingMCCFRSolver (outcome_sampling_mccfr.cc:141-241,
                                       Baseline() == 0), same initial values, same mini-batch
                                       protocol through osg_mccfr_sample / osg_mccfr_iterate        */
  double epsilon;                /* solver 2 only: OutcomeSamplingMCCFRSolver's exploration
                                    (outcome_sampling_mccfr.h:43 kDefaultEpsilon = 0.6)          */
  int32_t kernel;                /* 0 auto; 1 force the general level-synchronous kernel (k_cfr)
                                    even where the all-in-LDS small-tree kernel applies; 2 force the
                                    full-grid phase kernels (auto for trees > 65536 histories); 3 force
                                    the single-workgroup path kernel; 4 = auto's choice for trees that
                                    start with their chance deals and split into <= #CUs subtrees of
                                    <= 1024 histories (leduc_poker): one workgroup per deal subtree,
                                    one grid barrier per player pass; 5 = auto's choice for the big
                                    trees of that shape (3-player leduc_poker: 1.8 M histories): ONE
                                    cooperative launch, a workgroup per deal subtree of up to 8192
                                    histories, two grid barriers per player pass (alternating
                                    updates only)                                                 */
  int32_t replicas;              /* 0 or 1: one solver.  B > 1: B independent solvers of the same
                                    game advanced together, one workgroup each (CFR family, trees
                                    that fit LDS); select one with osg_cfr_select_replica         */
  int32_t random_initial_regrets;/* CFRSolverBase ctor (cfr.h:190-196): regrets start at
                                    0.001 * U[0,1) (cfr.cc:31,249-252) from the counter stream
                                    (seed, replica_offset + replica, infostate, action)           */
  uint64_t seed;
  int64_t replica_offset;        /* global index of replica 0 (sharding replicas over GPUs)       */
} osg_cfr_cfg;
/* Replaces CFRSolverBase::CFRSolverBase + InitializeInfostateNodes
 * (cfr.cc:191-261): expands the whole game tree level by level ON THE DEVICE
 * (legal-mask / apply / status kernels), numbers infostates, uploads the
 * level-ordered arrays and zero-initialised [I, Amax] fp64 tables. */
int osg_cfr_create(osg_ctx* ctx, const char* game_string, const osg_cfr_cfg* cfg, osg_cfr** out);
int osg_cfr_destroy(osg_cfr* s);
/* out[0..5] = histories, chance nodes, decision nodes, terminal nodes, infostates, Amax */
int osg_cfr_sizes(const osg_cfr* s, int64_t* out);
/* Back to iteration 0 with freshly initialised tables. */
int osg_cfr_reset(osg_cfr* s);
/* CFRSolverBase::EvaluateAndUpdatePolicy x iters (cfr.cc:263-282): one launch, all iterations. */
int osg_cfr_iterate(osg_cfr* s, int iters);
/* Number of EvaluateAndUpdatePolicy calls (or MCCFR mini-batches) done so far. */
int osg_cfr_iteration(const osg_cfr* s);
/* Diagnostic: the kernel family the solver's last iterate / sample call launched ("k_cfr_small<lds, owner>", "k_cfr_split",
 * "k_mccfr_resident_flat", "k_mccfr_resident<split 2>", ...; "" before the first launch or for the remaining forms).  The
 * parity tests and bench.py record it next to what they checked (no reference counterpart). */
const char* osg_cfr_last_kernel(const osg_cfr* s);
/* Diagnostic: the form the last policy evaluation took ("k_geval": a launch per level and phase, "k_geval_persist": one
 * persistent launch — opt-in, OSG_EVAL_PERSIST=1 —, "k_eval_jobs", "k_policy_eval"; "" before the first).  No reference counterpart. */
const char* osg_cfr_last_eval_kernel(const osg_cfr* s);
/* Number of replicas, and which one the table accessors / osg_cfr_evaluate_policy / upload act on. */
int osg_cfr_replicas(const osg_cfr* s);
int osg_cfr_select_replica(osg_cfr* s, int replica);
/* Restores the iteration counter of a deserialised solver (cfr.h:318-323 deserialisation ctor). */
int osg_cfr_set_iteration(osg_cfr* s, int iteration);
/* ExternalSamplingMCCFRSolver::FullUpdateAverage (external_sampling_mccfr.cc:188-231) — AverageType::kFull:
 * one full-tree pass adding weight * reach_probs[cur_player] * sigma(I)[a] to the cumulative policy of every
 * decision history, sigma = regret matching of the regrets as they are now.  weight = 1 after each
 * RunIteration's traversals is the reference; a mini-batch of T trajectories stands for T / P iterations. */
int osg_mccfr_full_average(osg_cfr* s, double weight);

/* AverageType of an external-sampling solver (external_sampling_mccfr.h:48): 0 kSimple (default), 1 kFull —
 * the traversals' sampled average-policy terms are dropped and osg_mccfr_iterate ends with
 * osg_mccfr_full_average(trajectories / P) when the batch holds at least one whole iteration. */
int osg_mccfr_set_average_type(osg_cfr* s, int average_type);

/* ONE UpdateRegrets(root, player, rng) (external_sampling_mccfr.cc:122-186) whose uniforms are
 * h_uniforms[0], h_uniforms[1], ... in visiting order instead of the counter stream — with the doubles
 * std::uniform_real_distribution<double>(0, 1) draws from the caller's std::mt19937 this IS the reference's
 * traversal, draw for draw (RunIteration(std::mt19937*), external_sampling_mccfr.h:63-100).  Fills the
 * delta tables like osg_mccfr_sample (fold with osg_mccfr_apply_deltas); *consumed = uniforms used. */
int osg_mccfr_sample_uniforms(osg_cfr* s, int player, const double* h_uniforms, int n, int32_t* consumed);

/* CFRBRSolver::EvaluateAndUpdatePolicy (cfr_br.cc:48-83) x iters on a CFRSolverBase table (solver 0,
 * no linear averaging, no RM+): every player's best response to the current policy
 * (TabularBestResponse, best_response.cc:194-227), one regret / average-policy pass per player with
 * the other players following their best responses, ApplyRegretMatching. */
int osg_cfr_br_iterate(osg_cfr* s, int iters);

/* ExternalSamplingMCCFRSolver::RunIteration (external_sampling_mccfr.cc:71-186,
 * AverageType::kSimple) for `trajectories` traverser passes (player = global
 * trajectory index mod P), mini-batched: every trajectory of one call reads the
 * tables as they were at the start of the call and adds its deltas atomically.
 * Tables start at kInitialTableValues = 1e-6 (external_sampling_mccfr.h:59). */
int osg_mccfr_iterate(osg_cfr* s, uint64_t seed, int64_t first_trajectory, int64_t trajectories);
/* The sampling half of osg_mccfr_iterate: zero the delta tables, run the traversals,
 * leave the regret / average-policy deltas in the delta tables (osg_mccfr_delta_ptrs)
 * WITHOUT folding them in.  Trajectory g (global index) draws from the counter stream
 * (seed, g) and updates player g mod P, so a batch can be split across GPUs by index.
 * External sampling: the draws below the traverser's first two nodes come from sub-streams
 * of (seed, g) — one per child subtree, in visiting order inside it — so that small
 * mini-batches can walk those subtrees on separate lanes (same sums as on one lane;
 * csrc/osg_cfr.hip es_stream, restated by the oracle's replay).  osg_mccfr_apply_deltas
 * leaves the delta tables zero. */
int osg_mccfr_sample(osg_cfr* s, uint64_t seed, int64_t first_trajectory, int64_t trajectories);
/* Device pointers to the [I, Amax] fp64 tables: regrets, cumulative policy,
 * current policy (for RCCL all-reduce by the caller, or inspection). */
int osg_cfr_table_ptrs(osg_cfr* s, double** d_regrets, double** d_cum_policy, double** d_cur_policy);
/* MCCFR mini-batch protocol for multi-GPU: deltas accumulate into separate
 * [I, Amax] buffers; the caller all-reduces them (RCCL) and then folds them in.
 * CONTRACT for the solver's own delta buffers (these and osg_mccfr_spare_delta_buffer's): the caller may write them
 * only between a sample and the apply that follows it (the in-place all-reduce does exactly that).  osg_mccfr_apply_deltas
 * leaves a buffer zero and the next osg_mccfr_sample relies on it — it skips its fill launch for a buffer the last fold
 * left clean — so anything written into the buffer after the apply is added to the next mini-batch.  Caller-owned
 * buffers (osg_mccfr_sample_into with memory of the caller) are always zero-filled first. */
int osg_mccfr_delta_ptrs(osg_cfr* s, double** d_regret_delta, double** d_policy_delta);
int osg_mccfr_apply_deltas(osg_cfr* s);
/* The same two halves on a CALLER's delta buffer d_delta = regret deltas [I, Amax] | average-policy deltas
 * [I, Amax] (2 * I * Amax device doubles, 8-byte aligned; NULL = the solver's own tables): with two such
 * buffers a host overlaps the all-reduce of mini-batch k with the traversals of mini-batch k + 1, which then
 * read the tables without k's deltas (stale by one mini-batch; ShardedMccfr(overlap=True) in
 * open_spiel_amd/distributed.py, ExternalSamplingMCCFRSolver::RunShardedMiniBatch(..., overlap) in the host
 * mirror).  The reference has no such schedule: its RunIteration is sequential (external_sampling_mccfr.cc:71-80). */
int osg_mccfr_sample_into(osg_cfr* s, uint64_t seed, int64_t first_trajectory, int64_t trajectories, double* d_delta);
int osg_mccfr_apply_deltas_from(osg_cfr* s, double* d_delta);
/* Two such buffers owned by the solver (allocated on first request, freed with it) for hosts that have no
 * device allocator of their own (which = 0 | 1). */
int osg_mccfr_spare_delta_buffer(osg_cfr* s, int which, double** d_delta);
/* Overwrite the [I, Amax] tables from host arrays (any may be NULL): restores a
 * checkpoint / CFRInfoStateValuesTable (cfr.cc:723-777 deserialisation target). */
int osg_cfr_upload_tables(osg_cfr* s, const double* h_regrets, const double* h_cum_policy,
                          const double* h_cur_policy);
/* CFRInfoStateValuesTable rows (cfr.h:42-104) to the host: arrays [I, Amax] (padding 0),
 * nact[I], legal[I, Amax] (padding -1), avg_policy per cfr.cc:104-125. Any may be NULL. */
int osg_cfr_tables(const osg_cfr* s, int32_t* nact, int32_t* legal, double* regrets,
                   double* cum_policy, double* cur_policy, double* avg_policy);
/* Policy evaluation on the flattened tree: algorithms::ExpectedReturns (expected_returns.cc:34-130),
 * TabularBestResponse values for every player (best_response.cc:194-227), NashConv and
 * Exploitability (tabular_exploitability.cc:30-89).  which_policy: 0 the tables' average policy
 * (cfr.cc:104-125), 1 their current policy, 2 the [I, Amax] host table h_policy (rows in the
 * solver's infostate order, osg_cfr_infostate_key).  expected_returns / best_response_values are
 * [P]; any output may be NULL. */
int osg_cfr_evaluate_policy(osg_cfr* s, int which_policy, const double* h_policy, double* expected_returns,
                            double* best_response_values, double* nash_conv, double* exploitability);

/* TabularBestResponse (best_response.h:38-130, best_response.cc:194-227) against the same policy choices as
 * osg_cfr_evaluate_policy: h_best_index [I] receives, for every infostate (of every player: each player responds
 * to the others playing the policy), the index among its legal actions of the best-response action (ties and
 * unreachable infostates: the first); best_response_values [P] (may be NULL) the responders' values. */
int osg_cfr_best_response(osg_cfr* s, int which_policy, const double* h_policy, int32_t* h_best_index,
                          double* best_response_values);
/* TabularBestResponse::Value(history) (best_response.h:127-128, best_response.cc:229-262) for EVERY history of the
 * flattened tree at once: h_history_values [H] (osg_cfr_sizes[0]) receives the value for `responder` of each history
 * when it best-responds from there on and the others follow the policy (choices as for osg_cfr_evaluate_policy). */
int osg_cfr_best_response_history_values(osg_cfr* s, int which_policy, const double* h_policy, int responder,
                                         double* h_history_values);
/* The flattened tree's edges: parent [H] (-1 at the root) and the action / chance outcome on the edge from the
 * parent [H] (-1 at the root); histories are numbered level by level.  Either may be NULL. */
int osg_cfr_tree_edges(const osg_cfr* s, int32_t* parent, int32_t* action);
/* InformationStateString() of infostate i (kuhn_poker.cc:109-166, leduc_poker.cc:198-239).
 * Returns the length (excluding NUL), or <0. */
int osg_cfr_infostate_key(const osg_cfr* s, int64_t i, char* buf, int cap);
/* The player who acts at infostate i (State::CurrentPlayer() of its histories), -1 for a bad index. */
int osg_cfr_infostate_player(const osg_cfr* s, int64_t i);

/* ---- multi-GPU exchange step ---------------------------------------------------
 * One process per GPU.  Units shard by global index with no data-path collective; the ONE exchange
 * on the path is external-sampling MCCFR's all-reduce(sum) of the regret / average-policy delta
 * tables per mini-batch (osg_mccfr_sample -> all-reduce of the 2 * I * Amax doubles starting at
 * osg_mccfr_delta_ptrs' regret pointer, the two tables being one allocation -> osg_mccfr_apply_deltas),
 * plus, for root-parallel search on a shared root, the children's visit / reward vectors.  The
 * reference has no distributed runtime (single-threaded C++); these entry points are what a C++
 * host of its shape binds instead of torch.distributed.  RCCL is loaded on first use (dlopen), so
 * single-GPU callers never need it.  The 128-byte id comes from rank 0 (osg_comm_unique_id) and
 * reaches the other ranks out of band (file, socket, MPI, launcher environment), as with
 * ncclGetUniqueId / ncclCommInitRank.  Collectives run in place on the context's stream. */
#define OSG_COMM_ID_BYTES 128
typedef struct osg_comm osg_comm;
int osg_comm_unique_id(void* id_out /* OSG_COMM_ID_BYTES */);
int osg_comm_create(osg_ctx* ctx, int rank, int world, const void* id /* OSG_COMM_ID_BYTES */, osg_comm** out);
int osg_comm_destroy(osg_comm* c);
/* The one-shot all-reduce for the path's latency-bound messages ("prefer a one-shot direct
 * all-reduce (each rank writes its buffer to all 7 peers, reduces locally)" for <= 45 KB): every rank owns a window
 * in its HBM that all peers map through hipIpc; one launch per call pushes the local buffer into every window over
 * xGMI, waits for the peers' pushes and sums the world's slots IN RANK ORDER — all ranks end with bit-identical
 * sums (a ring associates differently per rank).  No RCCL involved.
 *   1. every rank: osg_comm_oneshot_create(ctx, rank, world, max_doubles, &c)   max_doubles <= 32768, world <= 16
 *   2. every rank: osg_comm_oneshot_handle(c, h)  -> OSG_ONESHOT_HANDLE_BYTES bytes; all-gather them out of band
 *      (the channel the ncclUniqueId would travel on), rank order
 *   3. every rank: osg_comm_oneshot_connect(c, all_handles)
 *   4. osg_allreduce_sum_f64 / _i32 / _f64_begin + osg_allreduce_end as with the RCCL kind; osg_comm_destroy.
 * Ranks may share a device (two processes on one GPU map each other's windows just the same), which is how the
 * 1-GPU test box exercises it.  Failure is all-or-nothing per chunk and never silent: a peer that stays SILENT for
 * OSG_ONESHOT_TIMEOUT_MS (default 120 000; the clock restarts with every chunk that arrives, so a late rank is waited
 * for; 0 = no bound) makes the waiting workgroup POISON its chunk of the caller's buffer (NaN / INT32_MIN) instead of
 * leaving local values beside reduced ones, and raises a sticky error that osg_comm_check and every later call on
 * the communicator report.  What the bound is and is not: it bounds the SILENCE of a peer (the clock restarts with
 * every chunk that arrives), so a call can take up to world x OSG_ONESHOT_TIMEOUT_MS before it gives up; the poison and
 * the sticky error are LOCAL to the rank whose wait ran out — peers that received every chunk reduce normally and do
 * not learn of it from the collective — and the int32 poison INT32_MIN is a value, it does not propagate like a NaN.
 * Hence the caller's rule: every rank calls osg_comm_check after the last collective and the ranks AGREE on the result
 * (one more exchange on the channel the handles travelled on) before any of them folds / publishes what the
 * collective produced; open_spiel_amd/distributed.py ShardedMccfr.finish() does that over torch.distributed (a C++ host
 * calls Communicator::CheckHealth() and exchanges the verdict on the channel its handles travelled on).  The reference
 * has no counterpart (no distributed runtime). */
#define OSG_ONESHOT_HANDLE_BYTES 128
int osg_comm_oneshot_create(osg_ctx* ctx, int rank, int world, int64_t max_doubles, osg_comm** out);
int osg_comm_oneshot_handle(const osg_comm* c, void* handle_out /* OSG_ONESHOT_HANDLE_BYTES */);
int osg_comm_oneshot_connect(osg_comm* c, const void* handles /* world x OSG_ONESHOT_HANDLE_BYTES, rank order */);

/* Waits for the collectives issued so far (both streams) and reports a one-shot timeout, if any: call it before
 * trusting a buffer that went through the LAST collective of a job (a timeout is otherwise reported by the next call). */
int osg_comm_check(osg_comm* c);
int osg_comm_rank(const osg_comm* c);
int osg_comm_world(const osg_comm* c);
int osg_allreduce_sum_f64(osg_comm* c, double* d_buf, int64_t n);
int osg_allreduce_sum_i32(osg_comm* c, int32_t* d_buf, int64_t n);
/* The asynchronous form: _begin orders the collective after everything issued on the context's stream so far
 * and runs it on the communicator's own stream; kernels issued on the context's stream between _begin and
 * _end overlap it (they must not touch d_buf); _end makes the context's stream wait for the collective (no
 * host wait).  One collective in flight per communicator. */
int osg_allreduce_sum_f64_begin(osg_comm* c, double* d_buf, int64_t n);
int osg_allreduce_end(osg_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* OSG_ABI_H_ */
