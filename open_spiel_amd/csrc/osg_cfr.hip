// Tabular CFR / CFR+ (open_spiel/algorithms/cfr.{h,cc}) and external-sampling
// MCCFR (open_spiel/algorithms/external_sampling_mccfr.{h,cc}) on the device.
//
// The reference walks the game tree recursively, cloning a State per edge and
// looking infostates up by string (cfr.cc:331-408,443-469).  Here the tree is
// flattened ONCE, level by level, by the batched step kernels themselves
// (osg_legal_mask / osg_status_query / osg_batch_gather / osg_apply), and every
// iteration is arithmetic over flat arrays:
//
//   histories   level-ordered (BFS); children of a node are contiguous
//   per node    kind (chance / decision / terminal), actor, parent, first_child,
//               nchild, edge index within the parent, infostate id,
//               chance probability of the incoming edge, terminal returns [P]
//   infostates  [I, Amax] fp64 tables: cumulative regrets, cumulative policy,
//               current policy (CFRInfoStateValues, cfr.h:42-98); member
//               histories listed in the reference's DFS visiting order
//
// k_cfr runs a whole batch of iterations in ONE launch by ONE workgroup:
// top-down reach pass, bottom-up value pass (one __syncthreads per tree level),
// then one thread per infostate folds its member histories' regret / average
// policy terms in DFS order — the same additions in the same order as the
// reference's recursion, so the tables are bit-comparable with the CPU oracle
// (both sides are compiled with -ffp-contract=off).  Small trees (kuhn) keep
// reach / values / tables in LDS; larger ones (leduc: 9 457 histories) keep
// them in global memory, which is L2-resident at these sizes.
//
// k_mccfr runs one external-sampling traversal per thread with an explicit
// stack over the same flat tree; regret / average-policy deltas are accumulated
// per workgroup in LDS and flushed with fp64 atomics, so that a multi-GPU job
// can all-reduce the [I, Amax] delta tables once per mini-batch (RCCL) before
// every rank folds them in.
//
// Round 6: the kernel families live in their own translation units (osg_cfr_internal.h has the map); this file keeps
// the tree builder, the tables, and the entry points that dispatch to them.
#include "osg_cfr_internal.h"

namespace {

// ---------------------------------------------------------------------------
// InformationStateString of the acting player, from the packed state words
// (kuhn_poker.cc:109-166, leduc_poker.cc:198-239).
// ---------------------------------------------------------------------------
std::string kuhn_key(const Kuhn::Params& p, uint64_t word, int player) {
  Kuhn::State s{word};
  std::string r = Kuhn::len(s) > player ? std::to_string(Kuhn::card(s, player)) : std::string();  // not dealt yet: ""
  const int n = Kuhn::nact(p, s);
  for (int j = 0; j < n; ++j) r.push_back(((Kuhn::bets(s) >> j) & 1u) ? 'b' : 'p');
  return r;
}

// The part LeducObserver::StringFrom writes for both recall types (leduc_poker.cc:198-226).  money_ is
// kStartingMoney - ante_ while the hand runs; at the end the pot has been paid out (pot_ = 0,
// money = 100 + returns).
template <class L>
std::string leduc_observer_prefix(const LeducParams& p, const typename L::State& s, int player) {
  const bool term = L::terminal(p, s);
  double ret[kMaxPlayers] = {0};
  if (term) L::returns(p, s, ret);
  const int hole = L::priv(s, player);  // kInvalidCard = -10000 before the deal (leduc_poker.h:63)
  std::string r = "[Observer: " + std::to_string(player) + "][Private: " +
                  std::to_string(hole == L::kNone ? -10000 : hole) + "]";
  r += "[Round " + std::to_string(s.round) + "][Player: " + std::to_string(s.cur) + "][Pot: " +
       std::to_string(term ? 0 : s.pot) + "][Money: ";
  for (int q = 0; q < p.players; ++q) {
    char num[32];
    snprintf(num, sizeof num, "%g", term ? 100.0 + ret[q] : 100.0 - L::ante(s, q));
    if (q) r += " ";
    r += num;
  }
  r += "]";
  if (s.pub != L::kNone) r += "[Public: " + std::to_string(s.pub) + "]";
  return r;
}
template <class L>
std::string leduc_key_of(const LeducParams& p, const typename L::State& s, int player) {
  std::string r = leduc_observer_prefix<L>(p, s, player);
  for (int round = 0; round < 2; ++round) {
    r += round == 0 ? "[Round1: " : "][Round2: ";
    for (int k = 0; k < L::seqlen(s, round); ++k) {
      if (k) r += " ";
      r += std::to_string(static_cast<unsigned>((L::seq(s, round) >> (2 * k)) & 3u));
    }
  }
  r += "]";
  return r;
}
std::string leduc_key(const LeducParams& p, uint64_t w0, uint64_t w1, int player) {
  return leduc_key_of<Leduc>(p, Leduc::unpack(w0, w1), player);
}
// the state of either record from its plane words (2 or 5)
std::string leduc_key_words(const GameSpec& spec, const uint64_t* w, int player) {
  if (spec.leduc_big) return leduc_key_of<LeducBig>(spec.leduc, LeducBig::unpack5(w[0], w[1], w[2], w[3], w[4]), player);
  return leduc_key(spec.leduc, w[0], w[1], player);
}


// Expands the game tree breadth-first with the batched State kernels.
int build_tree(osg_cfr* s, const char* game_string) {
  osg_ctx* ctx = s->ctx;
  const osg_game_desc& d = s->spec.desc;
  const int P = d.num_players, W = d.mask_words, C = std::max(d.max_chance_outcomes, 1);
  const int words = d.state_words;
  const int64_t kMaxHistories = 1 << 24;
  s->P = P;
  s->A = d.num_distinct_actions;
  if (s->A > 255) return set_error(OSG_ERR_UNSUPPORTED, "osg_cfr_create: more than 255 distinct actions");

  std::unordered_map<std::string, int> key_to_id;
  osg_batch* level = nullptr;
  int rc = osg_batch_create(ctx, game_string, 1, &level);
  if (rc) return rc;
  s->level_off.push_back(0);
  // per-node data pending for the level being expanded: parent + edge data were written when the
  // parent was expanded; this loop fills in what depends on the state itself.
  s->parent.push_back(-1);
  s->aidx.push_back(0);
  s->edge_action.push_back(-1);
  s->edge_prob.push_back(0.0);
  int64_t level_begin = 0;
  while (level) {
    const int64_t n = osg_batch_size(level);
    std::vector<uint32_t> mask(static_cast<size_t>(n) * W);
    std::vector<int8_t> cur(n);
    std::vector<uint8_t> term(n);
    std::vector<double> rets(static_cast<size_t>(n) * P), probs(static_cast<size_t>(n) * C);
    std::vector<uint64_t> raw(static_cast<size_t>(n) * words);
    if ((rc = osg_legal_mask(level, mask.data(), 1)) || (rc = osg_status_query(level, cur.data(), term.data(), rets.data(), 1)) ||
        (d.max_chance_outcomes > 0 && (rc = osg_chance_probs(level, probs.data(), 1))) ||
        (rc = osg_batch_download(level, raw.data()))) {
      osg_batch_destroy(level);
      return rc;
    }
    std::vector<int64_t> gather;
    std::vector<int32_t> actions;
    const int64_t next_begin = level_begin + n;
    for (int64_t i = 0; i < n; ++i) {
      const int64_t h = level_begin + i;
      (void)h;
      s->actor.push_back(cur[i] >= 0 ? cur[i] : -1);
      if (term[i]) {
        s->kind.push_back(kTerminalNode);
        s->nchild.push_back(0);
        s->first_child.push_back(0);
        s->info.push_back(-1);
        for (int q = 0; q < P; ++q) s->term_ret.push_back(rets[i * P + q]);
        ++s->n_terminal;
        continue;
      }
      for (int q = 0; q < P; ++q) s->term_ret.push_back(0.0);
      const bool chance = cur[i] == kChancePlayer;
      s->kind.push_back(chance ? kChanceNode : kDecisionNode);
      s->first_child.push_back(static_cast<int32_t>(next_begin + static_cast<int64_t>(gather.size())));
      std::vector<int32_t> acts;
      for (int a = 0; a < 32 * W; ++a)
        if ((mask[i * W + (a >> 5)] >> (a & 31)) & 1u) acts.push_back(a);
      s->nchild.push_back(static_cast<uint8_t>(acts.size()));
      for (size_t k = 0; k < acts.size(); ++k) {
        gather.push_back(i);
        actions.push_back(acts[k]);
        s->parent.push_back(static_cast<int32_t>(h));
        s->aidx.push_back(static_cast<uint8_t>(k));
        s->edge_action.push_back(acts[k]);
        s->edge_prob.push_back(chance ? probs[i * C + acts[k]] : 0.0);
      }
      if (chance) {
        s->info.push_back(-1);
        ++s->n_chance;
        continue;
      }
      ++s->n_decision;
      std::string key;
      if (d.game_kind == kKuhn) key = kuhn_key(s->spec.kuhn, raw[i], cur[i]);
      else key = leduc_key(s->spec.leduc, raw[i], raw[n + i], cur[i]);
      auto it = key_to_id.find(key);
      int id;
      if (it == key_to_id.end()) {  // InitializeInfostateNodes (cfr.cc:234-261)
        id = static_cast<int>(s->keys.size());
        key_to_id.emplace(key, id);
        s->keys.push_back(key);
        s->nact.push_back(static_cast<int32_t>(acts.size()));
        s->info_player.push_back(cur[i]);
        for (int a = 0; a < s->A; ++a) s->legal.push_back(a < static_cast<int>(acts.size()) ? acts[a] : -1);
      } else {
        id = it->second;
        if (s->nact[id] != static_cast<int32_t>(acts.size())) {
          osg_batch_destroy(level);
          return set_error(OSG_ERR_INVALID, "infostate with inconsistent legal actions: " + key);
        }
      }
      s->info.push_back(id);
    }
    s->max_level_width = std::max<int>(s->max_level_width, static_cast<int>(n));
    s->level_off.push_back(static_cast<int32_t>(next_begin));
    level_begin = next_begin;
    osg_batch* next = nullptr;
    if (!gather.empty()) {
      if (next_begin + static_cast<int64_t>(gather.size()) > kMaxHistories) {
        osg_batch_destroy(level);
        return set_error(OSG_ERR_UNSUPPORTED, "osg_cfr_create: game tree exceeds 2^24 histories");
      }
      if ((rc = osg_batch_create(ctx, game_string, static_cast<int64_t>(gather.size()), &next)) ||
          (rc = osg_batch_gather(next, level, gather.data(), 1)) ||
          (rc = osg_apply(next, actions.data(), 1, nullptr))) {
        osg_batch_destroy(level);
        if (next) osg_batch_destroy(next);
        return rc;
      }
    }
    osg_batch_destroy(level);
    level = next;
  }
  s->H = static_cast<int>(level_begin);
  s->D = static_cast<int>(s->level_off.size()) - 1;
  s->I = static_cast<int>(s->keys.size());
  // Narrow the table width to the widest decision node.
  int amax = 1;
  for (int v : s->nact) amax = std::max(amax, v);
  if (amax != s->A) {
    std::vector<int32_t> legal(static_cast<size_t>(s->I) * amax);
    for (int i = 0; i < s->I; ++i)
      for (int a = 0; a < amax; ++a) legal[i * amax + a] = s->legal[i * s->A + a];
    s->legal.swap(legal);
    s->A = amax;
  }
  // Member histories of every infostate in the reference's DFS visiting order.
  std::vector<int32_t> order;
  order.reserve(s->H);
  std::vector<int32_t> stack{0};
  while (!stack.empty()) {
    const int h = stack.back();
    stack.pop_back();
    order.push_back(h);
    for (int a = s->nchild[h] - 1; a >= 0; --a) stack.push_back(s->first_child[h] + a);
  }
  std::vector<std::vector<int32_t>> members(s->I);
  for (int h : order)
    if (s->kind[h] == kDecisionNode) members[s->info[h]].push_back(h);
  s->mem_off.push_back(0);
  for (int i = 0; i < s->I; ++i) {
    s->mem.insert(s->mem.end(), members[i].begin(), members[i].end());
    s->mem_off.push_back(static_cast<int32_t>(s->mem.size()));
  }
  {  // tree level of every history; infostates must not span levels for the level-synchronous best response
    std::vector<int32_t> level_of(s->H, 0);
    for (int l = 0; l < s->D; ++l)
      for (int h = s->level_off[l]; h < s->level_off[l + 1]; ++h) level_of[h] = l;
    s->info_level.assign(s->I, -1);
    s->mem_index.assign(s->H, -1);
    for (int i = 0; i < s->I; ++i)
      for (int m = s->mem_off[i]; m < s->mem_off[i + 1]; ++m) {
        const int h = s->mem[m];
        s->mem_index[h] = m;
        if (s->info_level[i] < 0) s->info_level[i] = level_of[h];
        else if (s->info_level[i] != level_of[h]) s->eval_ok = false;
      }
  }
  s->first_decision_level = 0;
  for (int l = 0; l < s->D; ++l) {   // the first level that holds a decision history
    bool any = false;
    for (int h = s->level_off[l]; h < s->level_off[l + 1]; ++h) any |= s->kind[h] == kDecisionNode;
    if (any) { s->first_decision_level = l; break; }
  }
  // Root path of every decision history, root-to-leaf: one entry per ancestor edge =
  // (reach slot of the ancestor's actor, where to read the edge probability).
  s->path_off.push_back(0);
  for (int32_t h : s->mem) {
    std::vector<int32_t> rev;
    for (int32_t v = h; s->parent[v] >= 0; v = s->parent[v]) {
      const int32_t par = s->parent[v];
      const bool chance = s->kind[par] == kChanceNode;
      const int slot = chance ? s->P : s->actor[par];
      const int32_t idx = chance ? v : s->info[par] * s->A + s->aidx[v];
      rev.push_back((slot << 24) | ((chance ? 1 : 0) << 23) | idx);
    }
    int decisions = 0;
    for (int32_t code : rev) decisions += ((code >> 23) & 1) ? 0 : 1;
    s->max_path_decisions = std::max(s->max_path_decisions, decisions);
    s->path.insert(s->path.end(), rev.rbegin(), rev.rend());
    s->path_off.push_back(static_cast<int32_t>(s->path.size()));
  }
  return OSG_OK;
}

int init_tables(osg_cfr* s) {
  const int IA = s->I * s->A;
  hipStream_t st = s->ctx->stream;
  const size_t stride = s->replica_stride();
  std::vector<double> host(static_cast<size_t>(s->B) * stride, 0.0);
  for (int r = 0; r < s->B; ++r) {
    double* regrets = host.data() + r * stride;
    double* cum = regrets + IA;
    double* cur = cum + IA;
    for (int i = 0; i < s->I; ++i) {
      const int n = s->nact[i];
      double sum_pos = 0.0;
      for (int a = 0; a < n; ++a) {
        double init = s->cfg.solver >= 1 ? kMccfrInit : 0.0;
        if (s->cfg.random_initial_regrets) {
          // CFRInfoStateValues(la, rng, kRandomInitialRegretsMagnitude = 0.001) (cfr.h:52-61, cfr.cc:31,249-252):
          // regret = magnitude * U[0, 1); the reference draws from mt19937 + absl::Uniform (stream
          // unpinned), here from the counter stream (seed, replica, infostate * Amax + action).
          Rng rng(s->cfg.seed, static_cast<uint64_t>(s->cfg.replica_offset + r), static_cast<uint64_t>(i) * s->A + a);
          init = 0.001 * rng.unit();
        }
        regrets[i * s->A + a] = init;
        cum[i * s->A + a] = s->cfg.solver >= 1 ? kMccfrInit : 0.0;
        if (init > 0) sum_pos += init;
      }
      for (int a = 0; a < n; ++a) {  // ApplyRegretMatching on the fresh row (uniform when all regrets are 0)
        const double rg = regrets[i * s->A + a];
        cur[i * s->A + a] = (s->cfg.random_initial_regrets && sum_pos > 0) ? (rg > 0 ? rg / sum_pos : 0.0) : 1.0 / n;
      }
    }
  }
  OSG_HIP(hipMemcpyAsync(s->d_tables, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice, st));
  OSG_HIP(hipStreamSynchronize(st));
  s->iteration = 0;
  return OSG_OK;
}

}  // namespace

namespace osg_cfr_impl {

// A grid barrier of k_cfr_split / k_cfr_sub that timed out (a hung device: the launches are cooperative) leaves the
// tables mixed: the solver refuses further work.  The kernels raise a pinned host word, read here without a copy or a
// wait — by every entry point that advances, reads or hands out the tables.
int cfr_sub_error(const osg_cfr* s) {
  if (s->h_sub_err && __atomic_load_n(s->h_sub_err, __ATOMIC_RELAXED) != 0)
    return set_error(OSG_ERR_HIP, "a grid barrier of the subtree CFR kernel timed out in an earlier launch (the launch is "
                                  "cooperative: a hung device, not contention); the tables are not usable — "
                                  "osg_cfr_cfg.kernel = 3 runs one workgroup");
  return OSG_OK;
}

}  // namespace osg_cfr_impl

extern "C" {

int osg_cfr_create(osg_ctx* ctx, const char* game_string, const osg_cfr_cfg* cfg, osg_cfr** out) {
  if (!ctx || !game_string || !cfg || !out) return set_error(OSG_ERR_INVALID, "osg_cfr_create: null argument");
  if (ctx->closed) return set_error(OSG_ERR_INVALID, "osg_cfr_create: the context was destroyed");
  osg_cfr* s = new osg_cfr;
  s->ctx = ctx;
  s->cfg = *cfg;
  int rc = parse_game(game_string, &s->spec);
  if (rc) { delete s; return rc; }
  if (s->spec.desc.game_kind != kKuhn && s->spec.desc.game_kind != kLeduc) {
    delete s;
    return set_error(OSG_ERR_UNSUPPORTED, "tabular CFR needs information-state strings: kuhn_poker and leduc_poker only");
  }
  if (s->spec.leduc_big) {   // (4-player leduc_poker has ~3e8 histories, 10 players beyond any table)
    delete s;
    return set_error(OSG_ERR_UNSUPPORTED, "tabular solvers: leduc_poker with up to 3 players (the trees of 4+ players do not fit a device)");
  }
  s->B = s->cfg.replicas > 0 ? s->cfg.replicas : 1;
  if (s->B > 1 && s->cfg.solver != 0) { delete s; return set_error(OSG_ERR_UNSUPPORTED, "replicas > 1: CFR family only"); }
  if (s->B > (1 << 16)) { delete s; return set_error(OSG_ERR_INVALID, "osg_cfr_cfg.replicas must be <= 65536"); }
  if (s->cfg.solver < 0 || s->cfg.solver > 2) { delete s; return set_error(OSG_ERR_INVALID, "osg_cfr_cfg.solver must be 0, 1 or 2"); }
  if (s->cfg.solver == 2 && !(s->cfg.epsilon > 0.0 && s->cfg.epsilon <= 1.0)) {
    delete s;
    return set_error(OSG_ERR_INVALID, "outcome sampling needs 0 < epsilon <= 1 (kDefaultEpsilon = 0.6)");
  }
  rc = build_tree(s, game_string);
  if (rc) { delete s; return rc; }
  if (s->cfg.solver == 2 && s->D > kMaxOsDepth) {
    delete s;
    return set_error(OSG_ERR_UNSUPPORTED, "outcome sampling: game tree deeper than 32 levels");
  }
  osg::ctx_retain(ctx);  // from here on the object dies through osg_cfr_destroy, which releases
  hipStream_t st = ctx->stream;
  if ((rc = upload(s->level_off, &s->d_level_off, st)) || (rc = upload(s->parent, &s->d_parent, st)) ||
      (rc = upload(s->first_child, &s->d_first_child, st)) || (rc = upload(s->info, &s->d_info, st)) ||
      (rc = upload(s->mem_off, &s->d_mem_off, st)) || (rc = upload(s->mem, &s->d_mem, st)) ||
      (rc = upload(s->nact, &s->d_nact, st)) || (rc = upload(s->kind, &s->d_kind, st)) ||
      (rc = upload(s->nchild, &s->d_nchild, st)) || (rc = upload(s->aidx, &s->d_aidx, st)) ||
      (rc = upload(s->actor, &s->d_actor, st)) || (rc = upload(s->info_player, &s->d_info_player, st)) ||
      (rc = upload(s->edge_prob, &s->d_edge_prob, st)) || (rc = upload(s->term_ret, &s->d_term_ret, st))) {
    osg_cfr_destroy(s);
    return rc;
  }
  const size_t IA = static_cast<size_t>(s->I) * s->A;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_tables), sizeof(double) * s->B * 5 * std::max<size_t>(IA, 1));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_reach), sizeof(double) * s->H * (s->P + 1));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_value), sizeof(double) * s->H * s->P);
  if (e != hipSuccess) { osg_cfr_destroy(s); return set_error(OSG_ERR_NOMEM, hipGetErrorString(e)); }
  // Whole solver state in LDS when it fits (gfx950: 160 KiB per workgroup; leave headroom).
  s->lds_bytes = sizeof(double) * (static_cast<size_t>(s->H) * (2 * s->P + 1) + 3 * IA);
  s->lds_resident = s->lds_bytes <= 96 * 1024;
  {  // the all-in-LDS kernel for small trees
    const size_t M = s->mem.size();
    const size_t doubles = static_cast<size_t>(s->H) * s->P + s->H + 3 * IA + 2 * M * s->A;
    const size_t ints = 3 * static_cast<size_t>(s->H) + M + (s->I + 1) + (M + 1) + s->path.size() + 2 * s->I + M + (s->D + 1);
    s->small_lds_bytes = doubles * 8 + ints * 4;
    const bool index_fits = static_cast<size_t>(s->H) < (1u << 23) && IA < (1u << 23);
    s->small_tree = index_fits && s->small_lds_bytes <= 64 * 1024 && s->P <= kMaxPlayers;
    s->path_kernel = index_fits && s->P <= kMaxPlayers;
    s->meta32.resize(s->H);
    for (int h = 0; h < s->H; ++h) s->meta32[h] = s->kind[h] | (s->nchild[h] << 2) | ((s->actor[h] + 1) << 10);
    s->info_player32.assign(s->info_player.begin(), s->info_player.end());
    if ((rc = upload(s->path_off, &s->d_path_off, st)) || (rc = upload(s->path, &s->d_path, st)) ||
        (rc = upload(s->info_level, &s->d_info_level, st)) || (rc = upload(s->mem_index, &s->d_mem_index, st)) ||
        (rc = upload(s->meta32, &s->d_meta32, st)) || (rc = upload(s->info_player32, &s->d_info_player32, st))) {
      osg_cfr_destroy(s);
      return rc;
    }
    const size_t eval_doubles = static_cast<size_t>(s->H) * (s->P + 1) + M + 2 * s->P + IA;
    e = hipMalloc(reinterpret_cast<void**>(&s->d_eval), sizeof(double) * eval_doubles);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_best), sizeof(int32_t) * std::max(s->I, 1));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_skip), sizeof(int32_t) * std::max<size_t>(M, 1));
    if (e == hipSuccess)
      e = hipMalloc(reinterpret_cast<void**>(&s->d_node_delta), sizeof(double) * 2 * std::max<size_t>(M * s->A, 1));
    if (e != hipSuccess) { osg_cfr_destroy(s); return set_error(OSG_ERR_NOMEM, hipGetErrorString(e)); }
    if (!index_fits) s->eval_ok = false;
  }
  cfr_small_prepare(s);   // (the LDS caps of the one-workgroup kernels: osg_cfr_small.hip)
  if (s->B > 1 && !s->small_tree) {
    osg_cfr_destroy(s);
    return set_error(OSG_ERR_UNSUPPORTED, "replicas > 1 need the all-in-LDS kernel (tree too large)");
  }
  rc = build_resident_tree(s);
  if (rc) { osg_cfr_destroy(s); return rc; }
  if (hipHostMalloc(reinterpret_cast<void**>(&s->h_sub_err), sizeof(unsigned int), hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    osg_cfr_destroy(s);
    return set_error(OSG_ERR_NOMEM, "osg_cfr_create: pinned error word");
  }
  *s->h_sub_err = 0;
  rc = build_split(s);
  if (rc == OSG_OK) rc = build_sub(s);
  if (rc == OSG_OK) rc = build_eval_jobs(s);
  if (rc == OSG_OK && hipHostMalloc(reinterpret_cast<void**>(&s->h_eval_out), sizeof(double) * (2 * s->P + 1), hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    rc = set_error(OSG_ERR_NOMEM, "osg_cfr_create: pinned result buffer");
  }
  if (rc) { osg_cfr_destroy(s); return rc; }
  rc = init_tables(s);
  if (rc) { osg_cfr_destroy(s); return rc; }
  *out = s;
  return OSG_OK;
}

int osg_cfr_destroy(osg_cfr* s) {
  if (!s) return OSG_OK;
  (void)hipStreamSynchronize(s->ctx->stream);
  void* ptrs[] = {s->d_level_off, s->d_parent, s->d_first_child, s->d_info, s->d_mem_off, s->d_mem, s->d_nact,
                  s->d_kind, s->d_nchild, s->d_aidx, s->d_actor, s->d_info_player, s->d_edge_prob, s->d_term_ret,
                  s->d_tables, s->d_reach, s->d_value, s->d_path_off, s->d_path, s->d_info_level, s->d_mem_index,
                  s->d_best, s->d_eval, s->d_eval_ev, s->d_eval_level_info, s->d_eval_level_off, s->d_geval_bar, s->d_meta32, s->d_info_player32, s->d_skip, s->d_node_delta, s->d_rec,
                  s->d_uret, s->d_uprob, s->d_spare_delta[0], s->d_spare_delta[1], s->d_split_nloc, s->d_split_desc,
                  s->d_split_fc, s->d_split_row, s->d_split_glob, s->d_split_mem_m, s->d_split_mem_hloc, s->d_split_info,
                  s->d_split_terms, s->d_split_bar, s->d_sub_nloc, s->d_sub_desc, s->d_sub_fc, s->d_sub_aux, s->d_sub_mem_off,
                  s->d_sub_info_off, s->d_sub_info_list, s->d_sub_bar, s->d_sub_ndec, s->d_sub_dec_row, s->d_sub_rec,
                  s->d_sub_stamps, s->d_mccfr_stamps, s->d_sub_recbuf, s->d_sub_chance_prob, s->d_sub_term_val, s->d_sub_dec_off, s->d_sub_fold_info, s->d_sub_fold_off, s->d_sub_nroot, s->d_sub_root_loc, s->d_sub_root_idx, s->d_sub_upper_rec, s->d_sub_root_value,
                  s->d_jobs_job, s->d_jobs_level, s->d_jobs_desc, s->d_jobs_fc, s->d_jobs_row, s->d_jobs_glob, s->d_jobs_info,
                  s->d_jobs_mem, s->d_jobs_deal, s->d_jobs_ticket};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (s->h_sub_err) (void)hipHostFree(s->h_sub_err);
  if (s->h_eval_out) (void)hipHostFree(s->h_eval_out);
  osg::ctx_release(s->ctx);
  delete s;
  return OSG_OK;
}

int osg_cfr_sizes(const osg_cfr* s, int64_t* out) {
  if (!s || !out) return set_error(OSG_ERR_INVALID, "osg_cfr_sizes: null argument");
  out[0] = s->H; out[1] = s->n_chance; out[2] = s->n_decision; out[3] = s->n_terminal; out[4] = s->I; out[5] = s->A;
  return OSG_OK;
}

int osg_cfr_reset(osg_cfr* s) { return init_tables(s); }

int osg_cfr_iterate(osg_cfr* s, int iters) {
  if (!s || iters < 0) return set_error(OSG_ERR_INVALID, "osg_cfr_iterate: bad argument");
  if (iters == 0) return OSG_OK;
  if (int rc = cfr_sub_error(s)) return rc;
  int threads = ((s->max_level_width + 63) / 64) * 64;
  threads = std::max(64, std::min(threads, 1024));
  Tables tb{s->regrets(), s->cum(), s->cur()};
  if (s->B > 1) {  // all replicas advance together: workgroup b works on replica b's tables
    double* base0 = s->replica_base(0);
    tb = Tables{base0, base0 + static_cast<size_t>(s->I) * s->A, base0 + 2 * static_cast<size_t>(s->I) * s->A};
  }
  const unsigned grid_b = static_cast<unsigned>(s->B);
  // Trees far beyond one workgroup: full-grid launches per phase (osg_cfr_cfg.kernel == 2 forces it).
  const bool grid_path = s->path_kernel && s->B == 1 && (s->cfg.kernel == 2 || (s->cfg.kernel == 0 && s->H > 65536));
  // Trees too big for one workgroup: the persistent cooperative launch with a workgroup per deal subtree where the
  // tree has that shape (kernel == 5 forces it, 2 forces the per-phase launches), else a launch per phase.
  const bool sub_path = s->sub_ok && s->B == 1 && (s->cfg.kernel == 5 || (s->cfg.kernel == 0 && grid_path));
  if (sub_path) return cfr_sub_iterate(s, tb, iters);     // osg_cfr_sub.hip
  if (grid_path) return cfr_grid_iterate(s, tb, iters);   // osg_cfr_sub.hip
  if (s->split_ok && (s->cfg.kernel == 0 || s->cfg.kernel == 4)) {
    // one workgroup per deal subtree, one grid barrier per player pass (k_cfr_split)
    const int M = static_cast<int>(s->mem.size());
    SmallTree stree{s->d_path_off, s->d_path, M, static_cast<int>(s->path.size())};
    SplitTree sp{s->split_G, s->split_L, s->split_NL, s->split_NM, s->split_NI, s->d_split_nloc, s->d_split_desc,
                 s->d_split_fc, s->d_split_row, s->d_split_glob, s->d_split_mem_m, s->d_split_mem_hloc, s->d_split_info,
                 s->d_split_terms, s->d_split_bar, s->h_sub_err};
    const int passes = s->cfg.alternating_updates ? s->P : 1;
    const int per_launch = std::max(1, (1 << 30) / std::max(1, passes * s->split_G));  // the arrival counter is 32 bits
    for (int done = 0; done < iters; done += per_launch) {
      const int rc = launch_split(s, stree, sp, tb, std::min(per_launch, iters - done), s->iteration + done, s->cfg, false);
      if (rc) return rc;
    }
    OSG_HIP(hipGetLastError());
    s->iteration += iters;
    s->last_kernel = "k_cfr_split";
    return OSG_OK;
  }
  if (int rc = cfr_small_iterate(s, tb, iters, threads, grid_b)) return rc;   // osg_cfr_small.hip
  s->iteration += iters;
  return OSG_OK;
}

int osg_cfr_br_iterate(osg_cfr* s, int iters) {
  if (!s || iters < 0) return set_error(OSG_ERR_INVALID, "osg_cfr_br_iterate: bad argument");
  if (s->cfg.solver != 0) return set_error(OSG_ERR_INVALID, "osg_cfr_br_iterate: needs a CFRSolverBase table (solver 0)");
  if (s->cfg.linear_averaging || s->cfg.regret_matching_plus)
    return set_error(OSG_ERR_INVALID, "osg_cfr_br_iterate: CFRBRSolver is plain CFR (cfr_br.cc:23-29)");
  if (s->B != 1) return set_error(OSG_ERR_UNSUPPORTED, "osg_cfr_br_iterate: one solver per object");
  if (!s->eval_ok) return set_error(OSG_ERR_UNSUPPORTED, "an information state spans several tree levels");
  if (int rc = cfr_sub_error(s)) return rc;
  const size_t M = s->mem.size();
  EvalArrays ea;
  ea.path_off = s->d_path_off; ea.path = s->d_path; ea.info_level = s->d_info_level; ea.mem_index = s->d_mem_index;
  ea.M = static_cast<int>(M);
  ea.value = s->d_eval;
  ea.brv = ea.value + static_cast<size_t>(s->H) * s->P;
  ea.cf = ea.brv + s->H;
  ea.out = ea.cf + M;
  ea.best = s->d_best;
  int threads = ((s->max_level_width + 63) / 64) * 64;
  threads = std::max(64, std::min(threads, 1024));
  osg_cfr_cfg cfg = s->cfg;
  cfg.alternating_updates = 0;
  Tables tb{s->regrets(), s->cum(), s->cur()};
  // every player's best response to the current policy (cfr_br.cc:55-68), then one regret / average-policy
  // pass per player against the others' best responses (cfr_br.cc:70-81) and ApplyRegretMatching
  const bool jobs = s->jobs_ok && OSG_EVAL_JOBS_ENABLED();
  const bool split = s->split_ok && s->split_br_ok && (s->cfg.kernel == 0 || s->cfg.kernel == 4);
  SmallTree stree{s->d_path_off, s->d_path, static_cast<int>(M), static_cast<int>(s->path.size())};
  SplitTree sp{s->split_G, s->split_L, s->split_NL, s->split_NM, s->split_NI, s->d_split_nloc, s->d_split_desc,
               s->d_split_fc, s->d_split_row, s->d_split_glob, s->d_split_mem_m, s->d_split_mem_hloc, s->d_split_info,
               s->d_split_terms, s->d_split_bar, s->h_sub_err};
  if (!split && !jobs && eval_takes_the_grid(s) && s->path_kernel) {   // large trees (osg_cfr_sub.hip)
    // the P passes in ONE persistent launch per iteration (kernel == 2 keeps a launch per phase: the cross-check)
    if (s->sub_ok && s->sub_br_ok && s->cfg.kernel != 2) return cfr_sub_br_iterate(s, tb, ea, cfg, iters);
    return cfr_grid_br_iterate(s, tb, ea, cfg, iters);
  }
  for (int it = 0; it < iters; ++it) {
    if (int rc = cfr_best_responses_to_current(s, ea, threads, jobs)) return rc;   // osg_cfr_eval.hip
    if (split) {
      const int rc = launch_split(s, stree, sp, tb, 1, s->iteration, cfg, true);
      if (rc) return rc;
    } else {
      cfr_general_br_pass(s, tb, threads, cfg);   // osg_cfr_small.hip
    }
    ++s->iteration;
  }
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

int osg_cfr_table_ptrs(osg_cfr* s, double** d_regrets, double** d_cum_policy, double** d_cur_policy) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_cfr_table_ptrs: null argument");
  if (int rc = cfr_sub_error(s)) return rc;
  if (d_regrets) *d_regrets = s->regrets();
  if (d_cum_policy) *d_cum_policy = s->cum();
  if (d_cur_policy) *d_cur_policy = s->cur();
  return OSG_OK;
}

int osg_cfr_upload_tables(osg_cfr* s, const double* h_regrets, const double* h_cum_policy, const double* h_cur_policy) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_cfr_upload_tables: null argument");
  const size_t bytes = sizeof(double) * s->I * s->A;
  hipStream_t st = s->ctx->stream;
  if (h_regrets) OSG_HIP(hipMemcpyAsync(s->regrets(), h_regrets, bytes, hipMemcpyHostToDevice, st));
  if (h_cum_policy) OSG_HIP(hipMemcpyAsync(s->cum(), h_cum_policy, bytes, hipMemcpyHostToDevice, st));
  if (h_cur_policy) OSG_HIP(hipMemcpyAsync(s->cur(), h_cur_policy, bytes, hipMemcpyHostToDevice, st));
  OSG_HIP(hipStreamSynchronize(st));
  return OSG_OK;
}

int osg_cfr_tables(const osg_cfr* s, int32_t* nact, int32_t* legal, double* regrets, double* cum_policy,
                   double* cur_policy, double* avg_policy) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_cfr_tables: null argument");
  const size_t IA = static_cast<size_t>(s->I) * s->A, bytes = sizeof(double) * IA;
  hipStream_t st = s->ctx->stream;
  if (nact) memcpy(nact, s->nact.data(), sizeof(int32_t) * s->I);
  if (legal) memcpy(legal, s->legal.data(), sizeof(int32_t) * IA);
  std::vector<double> cum(IA);
  if (regrets) OSG_HIP(hipMemcpyAsync(regrets, s->regrets(), bytes, hipMemcpyDeviceToHost, st));
  if (cur_policy) OSG_HIP(hipMemcpyAsync(cur_policy, s->cur(), bytes, hipMemcpyDeviceToHost, st));
  OSG_HIP(hipMemcpyAsync(cum.data(), s->cum(), bytes, hipMemcpyDeviceToHost, st));
  unsigned int split_error = 0;
  if (s->split_ok) OSG_HIP(hipMemcpyAsync(&split_error, s->d_split_bar + 2, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
  OSG_HIP(hipStreamSynchronize(st));
  if (split_error)
    return set_error(OSG_ERR_HIP, "k_cfr_split: a grid barrier timed out after seconds (the launch is cooperative: this is a hung "
                                  "device, not contention); the tables are not usable — osg_cfr_cfg.kernel = 3 runs one workgroup");
  if (cum_policy) memcpy(cum_policy, cum.data(), bytes);
  if (avg_policy) {  // CFRAveragePolicy::GetStatePolicyFromInformationStateValues (cfr.cc:104-125)
    for (int i = 0; i < s->I; ++i) {
      const int n = s->nact[i];
      double sum = 0.0;
      for (int a = 0; a < n; ++a) sum += cum[i * s->A + a];
      for (int a = 0; a < s->A; ++a) {
        if (a >= n) avg_policy[i * s->A + a] = 0.0;
        else avg_policy[i * s->A + a] = sum == 0.0 ? 1. / n : cum[i * s->A + a] / sum;
      }
    }
  }
  return OSG_OK;
}

int osg_cfr_infostate_key(const osg_cfr* s, int64_t i, char* buf, int cap) {
  if (!s || i < 0 || i >= s->I || !buf || cap <= 0) return set_error(OSG_ERR_INVALID, "osg_cfr_infostate_key: bad argument");
  const std::string& k = s->keys[i];
  if (static_cast<int>(k.size()) + 1 > cap) return set_error(OSG_ERR_INVALID, "osg_cfr_infostate_key: buffer too small");
  memcpy(buf, k.c_str(), k.size() + 1);
  return static_cast<int>(k.size());
}

int osg_cfr_iteration(const osg_cfr* s) { return s ? s->iteration : 0; }
const char* osg_cfr_last_kernel(const osg_cfr* s) { return s ? s->last_kernel : ""; }
const char* osg_cfr_last_eval_kernel(const osg_cfr* s) { return s ? s->last_eval_kernel : ""; }
int osg_cfr_infostate_player(const osg_cfr* s, int64_t i) { return (s && i >= 0 && i < s->I) ? s->info_player[i] : -1; }
int osg_cfr_replicas(const osg_cfr* s) { return s ? s->B : 0; }
int osg_cfr_select_replica(osg_cfr* s, int replica) {
  if (!s || replica < 0 || replica >= s->B) return set_error(OSG_ERR_INVALID, "osg_cfr_select_replica: bad replica");
  s->selected = replica;
  return OSG_OK;
}
int osg_cfr_set_iteration(osg_cfr* s, int iteration) {
  if (!s || iteration < 0) return set_error(OSG_ERR_INVALID, "osg_cfr_set_iteration: bad argument");
  s->iteration = iteration;
  return OSG_OK;
}

int osg_cfr_tree_edges(const osg_cfr* s, int32_t* parent, int32_t* action) {
  if (!s) return set_error(OSG_ERR_INVALID, "osg_cfr_tree_edges: null solver");
  if (parent) memcpy(parent, s->parent.data(), sizeof(int32_t) * s->H);
  if (action) memcpy(action, s->edge_action.data(), sizeof(int32_t) * s->H);
  return OSG_OK;
}

int osg_information_state_string(const osg_batch* b, int64_t index, int player, char* buf, int cap) {
  if (!b || !buf || cap <= 0 || index < 0 || index >= b->n)
    return set_error(OSG_ERR_INVALID, "osg_information_state_string: bad argument");
  const osg_game_desc& d = b->spec.desc;
  if (d.game_kind != kKuhn && d.game_kind != kLeduc)
    return set_error(OSG_ERR_INVALID, "this game provides no information state string");
  if (player < 0 || player >= d.num_players) return set_error(OSG_ERR_INVALID, "player id out of range");
  uint64_t w[5] = {0, 0, 0, 0, 0};   // (leduc_poker with 4+ players: five plane words)
  const char* base = static_cast<const char*>(b->d_words);
  for (int k = 0; k < d.state_words; ++k)
    OSG_HIP(hipMemcpyAsync(&w[k], base + (static_cast<size_t>(k) * b->n + index) * sizeof(uint64_t), sizeof(uint64_t),
                           hipMemcpyDeviceToHost, b->ctx->stream));
  OSG_HIP(hipStreamSynchronize(b->ctx->stream));
  const std::string key = d.game_kind == kKuhn ? kuhn_key(b->spec.kuhn, w[0], player) : leduc_key_words(b->spec, w, player);
  if (static_cast<int>(key.size()) + 1 > cap) return set_error(OSG_ERR_INVALID, "buffer too small");
  memcpy(buf, key.c_str(), key.size() + 1);
  return static_cast<int>(key.size());
}

int osg_observation_string(const osg_batch* b, int64_t index, int player, char* buf, int cap) {
  if (!b || !buf || cap <= 0 || index < 0 || index >= b->n)
    return set_error(OSG_ERR_INVALID, "osg_observation_string: bad argument");
  const osg_game_desc& d = b->spec.desc;
  if (player < 0 || player >= d.num_players) return set_error(OSG_ERR_INVALID, "player id out of range");
  uint64_t w[4 * 12 + 1] = {0};   // (hex 19 x 19: 4 planes of 12 words and the meta word)
  const char* base = static_cast<const char*>(b->d_words);
  for (int k = 0; k < d.state_words; ++k)
    OSG_HIP(hipMemcpyAsync(&w[k], base + (static_cast<size_t>(k) * b->n + index) * d.state_word_bytes, d.state_word_bytes,
                           hipMemcpyDeviceToHost, b->ctx->stream));
  OSG_HIP(hipStreamSynchronize(b->ctx->stream));
  std::string out;
  switch (d.game_kind) {
    case kTtt: {  // tic_tac_toe.cc:163-175: rows joined by newlines
      const uint32_t x = static_cast<uint32_t>(w[0]) & 0x1FFu, o = (static_cast<uint32_t>(w[0]) >> 16) & 0x1FFu;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out += (x >> (3 * r + c) & 1u) ? "x" : ((o >> (3 * r + c) & 1u) ? "o" : ".");
        if (r < 2) out += "\n";
      }
      break;
    }
    case kC4: {  // connect_four.cc:212-222: top row first, every row ends with a newline
      const int R = b->spec.c4.rows, Cn = b->spec.c4.cols;
      osg_u128 x = b->spec.c4_std ? (w[0] & ((1ull << 56) - 1ull)) : w[0], o = w[1];
      if (b->spec.c4_wide) {   // two plane words per colour: x.lo, x.hi, o.lo, o.hi
        x = (static_cast<osg_u128>(w[1]) << 64) | w[0];
        o = (static_cast<osg_u128>(w[3]) << 64) | w[2];
      }
      for (int r = R - 1; r >= 0; --r) {
        for (int c = 0; c < Cn; ++c) {
          const int bit = c * (R + 1) + r;
          out += static_cast<uint32_t>(x >> bit & 1) ? "x" : (static_cast<uint32_t>(o >> bit & 1) ? "o" : ".");
        }
        out += "\n";
      }
      break;
    }
    case kHex: {  // hex.cc:341-359: one line per row, indented by the row number, a space after every cell
      const int NW = b->spec.hex_nw;
      int cols = 0, rows_unused = 0, cells = 0;
      hex_dims(b->spec, &rows_unused, &cols, &cells);
      auto bit = [&](int plane, int cell) { return (w[plane * NW + (cell >> 5)] >> (cell & 31)) & 1ull; };
      int line = 0;
      for (int i = 0; i < cells; ++i) {
        if (i && i % cols == 0) {
          out += "\n";
          out += std::string(++line, ' ');
        }
        const bool black = bit(0, i), white = bit(1, i);
        const int mag = 1 + 2 * static_cast<int>(bit(2, i)) + static_cast<int>(bit(3, i));  // plain, B, A, win
        if (!black && !white) out += ".";
        else if (!b->spec.hex_explicit) out += black ? "x" : "o";
        else out += black ? (mag == 1 ? "x" : mag == 2 ? "z" : mag == 3 ? "y" : "X")   // hex.cc:173-227
                          : (mag == 1 ? "o" : mag == 2 ? "q" : mag == 3 ? "p" : "O");
        out += " ";
      }
      break;
    }
    case kKuhn: {  // kuhn_poker.cc:109-166, default observer: own card, then every player's contribution
      const Kuhn::Params& kp = b->spec.kuhn;
      Kuhn::State st{w[0]};
      if (Kuhn::len(st) > player) {
        out += std::to_string(Kuhn::card(st, player));
        for (int q = 0; q < kp.players; ++q) out += std::to_string(Kuhn::did_bet(kp, st, q) ? 2 : 1);
      }
      break;
    }
    case kLeduc: {  // leduc_poker.cc:198-239, imperfect recall: pot contributions instead of the sequences
      const LeducParams& lp = b->spec.leduc;
      auto text = [&](auto tag, const auto& st) {
        using L = typename decltype(tag)::type;
        std::string r = leduc_observer_prefix<L>(lp, st, player) + "[Ante: ";
        for (int q = 0; q < lp.players; ++q) r += (q ? " " : "") + std::to_string(L::ante(st, q));
        return r + "]";
      };
      if (b->spec.leduc_big) out = text(TypeTag<LeducBig>{}, LeducBig::unpack5(w[0], w[1], w[2], w[3], w[4]));
      else out = text(TypeTag<Leduc>{}, Leduc::unpack(w[0], w[1]));
      break;
    }
    default: return set_error(OSG_ERR_INVALID, "bad game kind");
  }
  if (static_cast<int>(out.size()) + 1 > cap) return set_error(OSG_ERR_INVALID, "buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return static_cast<int>(out.size());
}

namespace {
// The packed words of one state, on the host.
int fetch_state_words(const osg_batch* b, int64_t index, uint64_t* w) {
  const osg_game_desc& d = b->spec.desc;
  const char* base = static_cast<const char*>(b->d_words);
  for (int k = 0; k < d.state_words; ++k)
    OSG_HIP(hipMemcpyAsync(&w[k], base + (static_cast<size_t>(k) * b->n + index) * d.state_word_bytes, d.state_word_bytes,
                           hipMemcpyDeviceToHost, b->ctx->stream));
  OSG_HIP(hipStreamSynchronize(b->ctx->stream));
  return OSG_OK;
}
const char* leduc_action_name(int a) { return a == 0 ? "Fold" : (a == 1 ? "Call" : "Raise"); }  // leduc_poker.cc:869-875
void hex_geometry(const GameSpec& spec, int* cols, int* rows, int* cells) { hex_dims(spec, rows, cols, cells); }
int return_string(const std::string& out, char* buf, int cap) {
  if (static_cast<int>(out.size()) + 1 > cap) return set_error(OSG_ERR_INVALID, "buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return static_cast<int>(out.size());
}
}  // namespace

int osg_state_string(const osg_batch* b, int64_t index, char* buf, int cap) {
  if (!b || !buf || cap <= 0 || index < 0 || index >= b->n)
    return set_error(OSG_ERR_INVALID, "osg_state_string: bad argument");
  const osg_game_desc& d = b->spec.desc;
  if (d.game_kind == kTtt || d.game_kind == kC4 || d.game_kind == kHex)
    return osg_observation_string(b, index, 0, buf, cap);  // ObservationString is ToString() in these games
  uint64_t w[4 * 12 + 1] = {0};   // (hex 19 x 19: 4 planes of 12 words and the meta word)
  int rc = fetch_state_words(b, index, w);
  if (rc) return rc;
  std::string out;
  if (d.game_kind == kKuhn) {  // kuhn_poker.cc:253-268: the dealt cards, then p / b
    const Kuhn::Params& kp = b->spec.kuhn;
    Kuhn::State st{w[0]};
    const int h = Kuhn::len(st);
    for (int i = 0; i < h && i < kp.players; ++i) out += (i ? " " : "") + std::to_string(Kuhn::card(st, i));
    if (h > kp.players) out += ' ';
    for (int j = 0; j < Kuhn::nact(kp, st); ++j) out.push_back(((Kuhn::bets(st) >> j) & 1u) ? 'b' : 'p');
  } else {  // leduc_poker.cc:463-496
    const LeducParams& lp = b->spec.leduc;
    auto text = [&](auto tag, const auto& st) {
      using L = typename decltype(tag)::type;
      const bool term = L::terminal(lp, st);
      double ret[kMaxPlayers] = {0};
      if (term) L::returns(lp, st, ret);
      const int P = lp.players;
      auto card = [](int c) { return std::to_string(c == L::kNone ? -10000 : c); };
      std::string r = "Round: " + std::to_string(st.round) + "\nPlayer: " + std::to_string(st.cur) + "\nPot: " +
                      std::to_string(term ? 0 : st.pot) + "\nMoney (player_0 player_1" + (P > 2 ? " [...]):" : "):");
      for (int q = 0; q < P; ++q) {
        char num[32];
        snprintf(num, sizeof num, "%g", term ? 100.0 + ret[q] : 100.0 - L::ante(st, q));
        r += std::string(" ") + num;
      }
      r += std::string("\nCards (public player_0 player_1") + (P > 2 ? " [...]): " : "): ") + card(st.pub) + " ";
      for (int q = 0; q < P; ++q) r += card(L::priv(st, q)) + " ";
      for (int round = 0; round < 2; ++round) {
        r += round == 0 ? "\nRound 1 sequence: " : "\nRound 2 sequence: ";
        for (int k = 0; k < L::seqlen(st, round); ++k)
          r += std::string(k ? ", " : "") + leduc_action_name(static_cast<int>((L::seq(st, round) >> (2 * k)) & 3u));
      }
      return r + "\n";
    };
    if (b->spec.leduc_big) out = text(TypeTag<LeducBig>{}, LeducBig::unpack5(w[0], w[1], w[2], w[3], w[4]));
    else out = text(TypeTag<Leduc>{}, Leduc::unpack(w[0], w[1]));
  }
  return return_string(out, buf, cap);
}

int osg_action_string(const osg_batch* b, int64_t index, int player, int32_t action, char* buf, int cap) {
  if (!b || !buf || cap <= 0 || index < 0 || index >= b->n)
    return set_error(OSG_ERR_INVALID, "osg_action_string: bad argument");
  const osg_game_desc& d = b->spec.desc;
  if (player < -1 || player >= d.num_players) return set_error(OSG_ERR_INVALID, "player id out of range");
  std::string out;
  switch (d.game_kind) {
    case kTtt:  // tic_tac_toe.cc:266-270
      out = std::string(player == 0 ? "x" : "o") + "(" + std::to_string(action / 3) + "," + std::to_string(action % 3) + ")";
      break;
    case kC4:  // connect_four.cc:158-161
      out = std::string(player == 0 ? "x" : "o") + std::to_string(action);
      break;
    case kHex: {  // hex.cc:295-314 (its "row" is the column letter)
      int cols, rows, cells;
      hex_geometry(b->spec, &cols, &rows, &cells);
      if (action == cells) { out = "swap"; break; }
      const int x = action % cols, y = action / cols;
      // Always the standard form, also for string_rep=explicit: hex.cc:301 tests `StringRep() == StringRep::kStandard`,
      // a value-initialised enum (= kStandard), not the state's string_rep(), so the reference never reaches its
      // explicit branch (:307-310).  Checked against the running reference (tests/test_oracle_vs_reference.py);
      // the board string (ToString, hex.cc:341-359) does honour string_rep.
      out = std::string(1, static_cast<char>('a' + x)) + std::to_string(y + 1);
      break;
    }
    case kKuhn:  // kuhn_poker.cc:244-251
      out = player < 0 ? "Deal:" + std::to_string(action) : std::string(action == 0 ? "Pass" : "Bet");
      break;
    case kLeduc:  // leduc_poker.cc:459-461
      out = player < 0 ? "Chance outcome:" + std::to_string(action) : std::string(leduc_action_name(action));
      break;
    default: return set_error(OSG_ERR_INVALID, "bad game kind");
  }
  return return_string(out, buf, cap);
}

}  // extern "C"

