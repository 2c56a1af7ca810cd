#!/usr/bin/env python3
"""Headline benchmark: batched connect_four LegalActions/ApplyAction env-steps/s.

Workload (BASELINE.json configs[1]): 2^20 parallel
connect_four states per GPU, state i = the initial position advanced by
hash(i) mod 36 uniformly random legal moves (never terminal), one uniformly
random legal action per state; seed 0x5EED.  One LAUNCH of the fused kernel
(legality check + ApplyAction + IsTerminal/CurrentPlayer/outcome + LegalActions
of the successor) passes once over the whole batch, out of place (src -> dst) so
that every launch does identical work.  A bench "step" = LAUNCHES_PER_STEP (100)
back-to-back launches, so that the timed region is long enough not to depend on
how few steps the caller asks for (a single launch lasts ~7 us); env-steps are
counted per launch.  Inputs are resident in HBM before the timed region.
N GPUs = N independent shards of 2^20 states (weak scaling, no collective on the
data path).

Beside the headline the line carries, all measured live in this process:
  roofline            the 2^20-state launch (36.7 MB: resident in the 256 MiB Infinity Cache, labelled so),
                      with the plain-copy time of the same bytes as `copy_ceiling`;
  roofline.dram_leg   the same kernel over 2^24 states (587 MB per launch, beyond every cache) with its own
                      copy ceiling: the DRAM-true figure;
  persistent          K = 32 random env steps per launch with the state in registers (osg_random_steps).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with no WORLD_SIZE in the environment re-executes itself as N ranks under
torch.distributed.run (one rank per GPU, RCCL; 127.0.0.1 rendezvous on a free port), so both command forms give
the same line.  At N > 1 the line keeps everything the N = 1 line has (rank 0 runs the roofline legs and the CPU
baseline while the other ranks wait on a host-side barrier) and adds per-rank rates, the collective's latency and the
strong-scaling efficiency of the sharded search against the same run's single-rank search of all roots.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STATES_PER_GPU = 1 << 20
DRAM_LEG_STATES = 1 << 24         # 587 MB per launch: beyond the 256 MiB Infinity Cache
LAUNCHES_PER_STEP = 100           # one bench "step" = this many back-to-back launches over the batch
SEED = 0x5EED
ALGO_BYTES_PER_STEP = 35          # 16 R + 16 W state, 1 action, 1 mask, 1 status
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec (MI355X_MICROARCH.md)


C4_DEPTH_MOD = 36                 # state i = hash(i) mod 36 random legal moves from the start
HEX_DEPTH_MOD = 40                # item 4: root i = hash(i) mod 40 random legal moves


def synth_batch(osa, torch, ctx, n, seed, index_offset):
    """SURVEY.md 8(d) item 2 on the device's counter stream (osg_synth_batch): state i = the initial position
    advanced by depth_i = draw(seed, index_offset + i) mod 36 uniformly random legal moves, re-drawn if the game
    ends earlier (never terminal), plus one uniformly random legal action.  The CPU reference regenerates the
    same batch from (seed, index range) — parity_check() below does, for the launches that were timed."""
    batch = osa.StateBatch(ctx, "connect_four", n)
    actions, _depth = batch.synth(seed, C4_DEPTH_MOD, index_offset=index_offset)
    return batch, actions


def parity_check(torch, src, dst, actions, mask, status, seed, index_offset, states, game="connect_four", depth_mod=None):
    """The timed launches against the CPU reference (the checker only): regenerate states
    [index_offset, index_offset + states) of the batch on the host threads with the GENUINE reference build
    (oracle/_ref, else the restatement) and compare, for every one of them, the source state, the action, and what
    the LAST timed launch left in dst / mask / status — successor state (as its ObservationTensor), successor
    legal mask, terminal flag, player to move, outcome.  Returns the record for the bench line; raises on any
    difference."""
    import numpy as np
    impl, kind = cpu_checker()
    threads = host_threads()
    t0 = time.perf_counter()
    rec = impl.Game(game).synth_batch(seed, states, depth_mod or C4_DEPTH_MOD, first=index_offset, threads=threads)
    cpu_s = time.perf_counter() - t0
    idx = torch.arange(states, device="cuda")

    def same(got, want, what):
        if not np.array_equal(got, want):
            bad = np.nonzero(np.asarray(got != want).reshape(states, -1).any(1))[0]
            raise AssertionError(f"parity_check: {what} differs from the {kind} at {bad.size} of {states} states "
                                 f"(first: {bad[:5].tolist()})")

    same(actions[:states].cpu().numpy().astype(np.int16), rec["action"], "action")
    for batch, sfx in ((src, "0"), (dst, "1")):
        part = batch if states == batch.n else batch.gather(idx)
        same(part.observation_tensor(0).to(torch.uint8).cpu().numpy(), rec["obs" + sfx], f"state{sfx}")
        same(part.legal_actions_mask_bits().cpu().numpy().view(np.uint32), rec["mask" + sfx], f"legal mask{sfx}")
        cur, term, rets = [t.cpu().numpy() for t in part.status()]
        same(cur, rec["cur" + sfx], f"current player{sfx}")
        same(term, rec["term" + sfx], f"terminal{sfx}")
        if sfx == "1":
            same(rets, rec["rets1"], "returns")
    st = status[:states].cpu().numpy()
    term1 = rec["term1"] != 0
    same((st & 0x80) != 0, term1, "status.terminal")
    same((st & 0x40) != 0, np.zeros(states, bool), "status.illegal")
    same(((st & 15).astype(np.int64) - 1)[~term1], rec["cur1"][~term1].astype(np.int64), "status.player")
    outcome = np.where(rec["rets1"][:, 0] > 0, 0, np.where(rec["rets1"][:, 0] < 0, 1, 2))
    same((st & 7)[term1], outcome[term1], "status.outcome")
    if mask is not None:   # (the hex step writes no mask row: ~occupied of the successor record, checked above as legal mask1)
        same(mask.reshape(-1)[:states].cpu().numpy(), (rec["mask1"][:, 0] & 0xFF).astype(np.uint8), "successor mask byte")
    return {"states": int(states), "against": kind, "cpu_seconds": cpu_s, "cpu_threads": threads,
            "what": "source state, action, successor state (ObservationTensor), successor legal mask, terminal flag, "
                    "player to move, outcome and Returns() of the last timed launch, every state compared"}


def host_threads():
    """Worker threads for the CPU baseline: the cores this process may actually use — the affinity mask,
    capped by the cgroup CPU quota (a container can see 256 CPUs and be allowed 16) and at 64."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_checker():
    """The CPU implementation timed beside the GPU: the GENUINE reference build
    (oracle/_ref/libspiel_ref.so — the reference's own .cc files compiled by oracle/Makefile.ref in
    the development container; the prebuilt file travels with the repo) when it loads, else the
    restatement.  Returns (binding, kind)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import reference_py
        if reference_py.available():
            reference_py.Game("connect_four")   # dlopen + game registry sanity
            return reference_py, "reference"
    except Exception as e:  # noqa: BLE001 - a baseline that cannot load must not sink the bench
        print(f"bench.py: genuine reference build unusable ({e}); timing the restatement", file=sys.stderr)
    import oracle_py
    oracle_py.build()
    return oracle_py, "port"


def cpu_baseline():
    """The reference's CPU path (see cpu_checker) timed on this box's host cores on a bounded
    sample of the same workload (~15-20 s of CPU work in total)."""
    impl, kind = cpu_checker()
    g = impl.Game("connect_four")
    pool = 1 << 14
    secs, units = g.bench_env_steps(SEED, pool, 200_000, 1)               # calibrate 1 thread
    secs1, units1 = g.bench_env_steps(SEED, pool, int(units / secs * 5), 1)   # ~5 s, 1 thread
    single = units1 / secs1
    threads = host_threads()
    secs, units = g.bench_env_steps(SEED, pool, 100_000 * threads, threads)   # calibrate N threads
    secs_n, units_n = g.bench_env_steps(SEED, pool, int(units / secs * 8), threads)  # ~8 s
    if secs_n < 4.0:  # the short calibration run under-estimated the rate: once more, scaled to ~8 s
        secs_n, units_n = g.bench_env_steps(SEED, pool, int(units_n / secs_n * 8), threads)
    rec = {
        "value": units_n / secs_n, "unit": "env-steps/s", "cores": threads, "kind": kind,
        "single_thread_value": single,
        "implementation": ("open_spiel/games/connect_four/connect_four.cc + spiel.cc compiled -O3 -DNDEBUG from the "
                           "reference sources (oracle/Makefile.ref; abseil / nlohmann stand-ins from oracle/ref_shim)"
                           if kind == "reference" else "oracle/ restatement (reference-shaped C++ port)"),
        "sample": (f"{units_n} connect_four env steps (Clone + LegalActions + ApplyAction + IsTerminal + "
                   f"Returns + CurrentPlayer per step) over a pool of {pool} seeded positions, "
                   f"{threads} threads, {secs_n:.1f} s; single thread {units1} steps in {secs1:.1f} s"),
        "sample_short": f"{units_n} env steps, {pool} seeded positions, {threads} threads, {secs_n:.1f} s",
    }
    if kind == "reference":  # the restatement beside it, single thread, ~2 s: how representative the port is
        import oracle_py
        oracle_py.build()
        s2, u2 = oracle_py.Game("connect_four").bench_env_steps(SEED, pool, int(single * 2), 1)
        rec["restatement_single_thread_value"] = u2 / s2
    return rec


def hex_roots(osa, torch, ctx, n, index_offset, seed=SEED, depth_mod=HEX_DEPTH_MOD):
    """hex(9) search roots on the counter stream: root i = the empty board advanced by
    draw(seed, index_offset + i) mod 40 uniformly random legal moves, never terminal (osg_synth_batch)."""
    b = osa.StateBatch(ctx, "hex(board_size=9)", n)
    b.synth(seed, depth_mod, index_offset=index_offset)
    return b


def mcts_parity_check(res, first, count, sims):
    """The timed search against the oracle's replay-mode MCTSBot (restatement; the checker only): the first `count`
    roots of this rank's shard, all `sims` simulations — best action, visit vector, reward vector."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    oracle_py.build()
    threads = host_threads()
    t0 = time.perf_counter()
    want = oracle_py.Game("hex(board_size=9)").synth_mcts_replay(SEED, count, HEX_DEPTH_MOD, 2.0, sims, 1, SEED, 2,
                                                                 first=first, threads=threads)
    cpu_s = time.perf_counter() - t0
    for key in ("best_action", "child_visits", "child_reward"):
        got = res[key][:count].cpu().numpy()
        if not np.array_equal(got, want[key]):
            bad = np.nonzero((got != want[key]).reshape(count, -1).any(1))[0]
            raise AssertionError(f"mcts_parity_check: {key} differs from the oracle replay at {bad.size} of {count} roots "
                                 f"(first: {bad[:5].tolist()})")
    return {"roots": int(count), "simulations_each": int(sims), "against": "port (replay-mode MCTSBot of the restatement)",
            "pin": "one step removed from mcts.cc: the genuine MCTSBot draws from absl::Uniform (mcts.cc:54), whose streams are "
                   "unspecified and absent here; the restatement's MCTSBot builds the genuine one's tree node for node under the "
                   "stand-in generator (tests/test_oracle_vs_reference.py), and replays the device's counter streams here",
            "cpu_seconds": cpu_s, "cpu_threads": threads,
            "what": "best action, child visit counts and child total rewards of the timed search, every checked root"}


def secondary_workloads(osa, torch, dist, ctx, rank, world, with_cpu, host_barrier=lambda: None,
                        gather_floats=lambda x: [float(x)]):
    """BASELINE.json configs 3-5 next to the headline: hex(9) MCTS sims/s (roots sharded over the
    ranks: strong scaling), kuhn CFR iterations/s (replicas only) and leduc ES-MCCFR trajectories/s
    (trajectories sharded, one RCCL all-reduce of the delta tables per mini-batch)."""
    import numpy as np
    from open_spiel_amd import distributed as osd
    out = {}

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- config 4: hex(board_size=9) MCTS, 2^16 roots x 1024 simulations, 1 rollout each ----
    total_roots, sims = 1 << 16, 1024
    first, count = osd.shard_range(total_roots, rank, world)
    roots = hex_roots(osa, torch, ctx, count, first)
    # warm-up with the same geometry: sizes the context's node pool and log table once
    roots.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=1, index_offset=first, layout=2)
    fence()
    t0 = time.perf_counter()
    res = roots.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=SEED, index_offset=first, layout=2)
    fence()
    dt_local = time.perf_counter() - t0
    done_local = float(res["root_stats"][:, 3].sum().item())
    per_rank_dt, per_rank_done = gather_floats(dt_local), gather_floats(done_local)
    dt, done = max(per_rank_dt), sum(per_rank_done)
    out["mcts"] = {"metric": "MCTS sims/sec", "value": done / dt, "unit": "sims/s", "seconds": dt,
                   "scaling": "strong",
                   "config": {"workload": "hex(board_size=9) MCTSBot(RandomRolloutEvaluator(1), uct_c=2, 1024 sims) "
                                          f"x 2^16 roots, {count} roots on rank 0, wave-per-root layout"}}
    if rank == 0 and with_cpu:
        try:
            out["mcts"]["parity"] = mcts_parity_check(res, first, min(count, 1024), sims)
            out["mcts"]["parity_checked_roots"] = out["mcts"]["parity"]["roots"]
        except AssertionError as e:
            out["mcts"]["parity"] = {"error": str(e)}
            out["mcts"]["parity_checked_roots"] = 0
    del roots, res
    if world > 1:
        # strong scaling against THIS run's one-GPU search of all 2^16 roots (rank 0 alone, the others wait)
        out["mcts"]["per_rank_sims_per_s"] = [d / t for d, t in zip(per_rank_done, per_rank_dt)]
        if rank == 0:
            full = hex_roots(osa, torch, ctx, total_roots, 0)
            full.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=1, index_offset=0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r1 = full.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=SEED, index_offset=0)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t0
            done1 = float(r1["root_stats"][:, 3].sum().item())
            out["mcts"]["single_rank_all_roots"] = {"value": done1 / dt1, "unit": "sims/s", "seconds": dt1,
                                                    "what": "the same 2^16 roots searched by rank 0 alone in this run"}
            out["mcts"]["strong_scaling_efficiency"] = (done / dt) / (world * done1 / dt1)
            del full, r1
        host_barrier()
    # Issue-rate view of the search kernel (it moves ~3.5 KB per simulation: far from a memory roofline): vector,
    # scalar and branch instructions per simulation from the committed counter profile, issue intervals from
    # profiles/r02_clock_probe.log.
    if rank == 0:
        mix = mcts_instruction_mix()
        if mix:
            simds = torch.cuda.get_device_properties(0).multi_processor_count * 4
            ns_per_sim = dt / (done / world / simds) * 1e9
            # issue intervals per SIMD measured by tools/clock_probe.hip with every SIMD saturated
            # (profiles/r02_clock_probe.log): a scalar-unit instruction every 1.833 ns, a simple vector
            # instruction every 1.03 ns (64-bit shifts, multiplies, fp64: about twice that)
            # (round 6: branches are NOT charged to the scalar ALU's issue slot any more — the search now runs faster
            # than (salu + branch) x 1.833 ns per simulation, 841 ns against 816 measured, so they cannot share it; the
            # branch unit is its own issue class)
            scalar_ns = mix["salu"] * 1.833
            vector_ns = mix["valu"] * 1.03
            out["mcts"]["roofline"] = {
                "bound": "scalar-unit instruction issue", "valu_per_sim": mix["valu"], "salu_per_sim": mix["salu"],
                "branch_per_sim": mix["branch"], "source": mix["source"], "source_current": source_is_current(mix["source"]),
                "ns_per_sim_per_simd": ns_per_sim,
                "scalar_issue_ns_per_sim": scalar_ns, "vector_issue_ns_per_sim_at_least": vector_ns,
                "frac_of_scalar_issue_bound": scalar_ns / ns_per_sim,
                "note": "the scalar ALU issues one instruction per 4 cycles per SIMD (1.833 ns measured with every wave slot "
                        "busy); fewer instructions per simulation on both pipes is the lever (round 2: 652 -> 387 scalar, "
                        "7.97e8 -> 1.12e9 simulations/s; round 6: 520 -> 442 vector — the UCT arg-max through an fp32 filter, "
                        "the threshold search on 32-bit words, the expansion from lane masks — 381 -> 376 scalar, 1.10e9 -> "
                        "1.256e9, profiles/r06zt_*); branches (83 per simulation) issue on their own port; "
                        "instruction counts are per simulation of an 8192-root search from the empty board"}

    # ---- config 3: kuhn_poker CFRSolver (full-tree regret / strategy update kernel) ----
    solver = osa.TabularSolver(ctx, "kuhn_poker")
    solver.evaluate_and_update_policy(100)
    fence()
    iters = 20000
    t0 = time.perf_counter()
    solver.evaluate_and_update_policy(iters)
    fence()
    dt = time.perf_counter() - t0
    out["cfr"] = {"metric": "CFR iterations/sec", "value": iters / dt, "unit": "iterations/s", "seconds": dt,
                  "scaling": "replicas only", "kernel": solver.last_kernel(),
                  "config": {"workload": f"kuhn_poker CFRSolver, {iters} EvaluateAndUpdatePolicy in one launch, "
                                         "58 histories / 12 infostates, LDS-resident"}}
    if rank == 0:
        rf = cfr_small_roofline(iters / dt)
        if rf:
            out["cfr"]["roofline"] = rf
    if rank == 0 and with_cpu:
        # what the timed launch left behind, against the CPU solver run for the same number of iterations (the checker
        # only: oracle/parity.py; the genuine reference build when it loads)
        try:
            impl, kind = cpu_checker()   # (also puts oracle/ on sys.path)
            import parity
            t = solver.tables()
            rec = parity.cfr_tables(impl, "kuhn_poker", "cfr", 100 + iters, t["keys"], t["nact"], t["regrets"],
                                    t["cum_policy"], t["avg_policy"])
            rec["against"] = ("oracle/_ref (the reference's CFRSolver, cfr.cc:263-469)" if kind == "reference"
                              else "oracle restatement of cfr.cc:263-469")
            rec["what"] = (f"regrets, cumulative policy and average policy of all 12 infostates after the 100 warm-up + {iters} "
                           "timed iterations, against the CPU solver run for the same 20 100 iterations")
            out["cfr"]["parity"] = rec
            out["cfr"]["parity_checked_iterations"] = rec["iterations"]
        except AssertionError as e:
            out["cfr"]["parity"] = {"error": str(e)}
            out["cfr"]["parity_checked_iterations"] = 0
    del solver
    # the same kernel with one workgroup per independent solver (random initial regrets): what the GPU does
    # with this config when asked for many solves at once
    replicas = 16384  # 16 wavefronts queued per SIMD; 4096 (4 per SIMD) gives 5.6e8, 16384 7.3e8, 32768 7.7e8
    many = osa.TabularSolver(ctx, "kuhn_poker", replicas=replicas, random_initial_regrets=True, seed=SEED,
                             replica_offset=rank * replicas)
    many.evaluate_and_update_policy(50)
    fence()
    rep_iters = 2000
    t0 = time.perf_counter()
    many.evaluate_and_update_policy(rep_iters)
    fence()
    dt = max_over_ranks(time.perf_counter() - t0)
    out["cfr"]["replicas"] = {"value": replicas * world * rep_iters / dt, "unit": "solver-iterations/s",
                              "seconds": dt, "scaling": "weak",
                              "workload": f"{replicas} independent kuhn_poker CFRSolver replicas per GPU (random initial "
                                          f"regrets), {rep_iters} iterations each, one workgroup per replica"}
    del many
    # the same solver on leduc_poker (9 457 histories, 936 infostates): one workgroup per deal subtree (k_cfr_split);
    # and on 3-player leduc (1.8 M histories), which runs the full-grid phase kernels
    try:
        big = osa.TabularSolver(ctx, "leduc_poker")
        big.evaluate_and_update_policy(50)
        fence()
        l_iters = 5000
        t0 = time.perf_counter()
        big.evaluate_and_update_policy(l_iters)
        fence()
        dt = time.perf_counter() - t0
        out["cfr"]["leduc"] = {"value": l_iters / dt, "unit": "iterations/s", "us_per_iteration": dt / l_iters * 1e6,
                               "nash_conv_after": big.nash_conv(),
                               "workload": f"leduc_poker CFRSolver, {l_iters} EvaluateAndUpdatePolicy in one launch: 30 workgroups "
                                           "(one per deal subtree, LDS-resident), one cross-workgroup exchange per player pass; "
                                           "tables bit-identical with the single-workgroup kernel"}
        del big
        if rank == 0:
            three = osa.TabularSolver(ctx, "leduc_poker(players=3)")
            # the first 2 iterations of the timed solver are what the CPU reference is asked to reproduce below (6.7 s per
            # reference iteration on this tree: the 55 that follow cannot be afforded)
            three.evaluate_and_update_policy(2)
            three_tables = three.tables() if with_cpu else None
            three.evaluate_and_update_policy(3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            three.evaluate_and_update_policy(50)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
            h3, p3 = three.num_histories, 3
            bytes_per_iteration = h3 * (4 + 1 + 2 * p3 * 8) * p3     # H x (parent 4 + action 1 + 2 P 8) per player pass
            out["cfr"]["leduc_3_players"] = {"value": 50 / dt3, "unit": "iterations/s", "us_per_iteration": dt3 / 50 * 1e6,
                                             "kernel": three.last_kernel(),
                                             "algorithmic_bytes_per_iteration": bytes_per_iteration,
                                             "frac_of_hbm_peak": bytes_per_iteration * 50 / dt3 / 1e9 / HBM_PEAK_GBS,
                                             "workload": "leduc_poker(players=3) CFRSolver, 1.83 M histories / 25 800 infostates: ONE "
                                                         "cooperative launch (k_cfr_sub, forest form): the 672 pieces below the 336 deal "
                                                         "roots packed into one bin per workgroup (256 bins of ~7 150 histories: values, "
                                                         "policy rows and chance probabilities in LDS, the rows resident between passes), two "
                                                         "two-level grid barriers per player pass with the next phase's static fetches in "
                                                         "their windows; tables bit-identical with the launch-per-phase kernels (1 870 it/s)",
                                             "note": "round 4: 3 260 it/s, round 5: 8 000-8 260, round 6: ~10 200 — a member's 64-byte record "
                                                     "written by its quad as one contiguous piece (the members phase 12 -> 5.7 us: profiles/r06h_*, "
                                                     "r06i_*), the two-member round per wavefront, the barrier's pollers on the top counter, the "
                                                     "bin's descriptors fetched once per launch (r06m_*, r06o_*, r06p_*).  Still bound by dependent "
                                                     "phases, not bytes: a pass of ~32 us is sweep ~10.5, members ~6, fold 6.5-8 and two barriers; "
                                                     "OSG_CFR_SUB_STAMPS prints the phase stamps and every workgroup's barrier arrivals"}
            del three
            if three_tables is not None:
                try:
                    impl, kind = cpu_checker()
                    import parity
                    t0 = time.perf_counter()
                    rec = parity.cfr_tables(impl, "leduc_poker(players=3)", "cfr", 2, three_tables["keys"], three_tables["nact"],
                                            three_tables["regrets"], three_tables["cum_policy"], three_tables["avg_policy"], rtol=1e-12)
                    rec["cpu_seconds"] = time.perf_counter() - t0
                    rec["against"] = ("oracle/_ref (the reference's CFRSolver, cfr.cc:263-469)" if kind == "reference"
                                      else "oracle restatement of cfr.cc:263-469")
                    rec["what"] = ("regrets, cumulative policy and average policy of all 25 800 infostates after the first 2 iterations "
                                   "of the timed solver (k_cfr_sub<forest>), 1 host thread")
                    out["cfr"]["leduc_3_players"]["parity"] = rec
                    out["cfr"]["leduc_3_players"]["parity_checked_iterations"] = rec["iterations"]
                    out["cfr"]["leduc_3_players"]["cpu_baseline"] = {
                        "value": 2 / rec["cpu_seconds"], "unit": "iterations/s", "cores": 1, "kind": kind,
                        "sample": f"2 CFRSolver iterations incl. the solver's construction, 1 thread, {rec['cpu_seconds']:.1f} s"}
                except AssertionError as e:
                    out["cfr"]["leduc_3_players"]["parity"] = {"error": str(e)[:500]}
                    out["cfr"]["leduc_3_players"]["parity_checked_iterations"] = 0
                del three_tables
    except Exception as e:  # noqa: BLE001 - a secondary figure must never cost the line
        out["cfr"]["leduc"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- config 5: leduc_poker external-sampling MCCFR, 2^24 trajectories, mini-batches of 2^20 ----
    solver = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    sharded = osd.ShardedMccfr(solver)
    sharded.run_minibatch(SEED, 1 << 12)
    fence()
    batch, nb = 1 << 20, 16
    # Timed in two bracketed segments, 15 mini-batches + 1, so that the LAST timed mini-batch can be checked: between the
    # segments (outside both clocks) the tables are copied on the device; afterwards the CPU replays that mini-batch's
    # 2^20 trajectories on the copied table and every cell of "table after - table before" is compared.
    t0 = time.perf_counter()
    for _ in range(nb - 1):
        sharded.run_minibatch(SEED, batch)
    fence()
    dt_a = time.perf_counter() - t0
    snap = [t.clone() for t in solver.device_tables()[:2]] if (rank == 0 and with_cpu) else None
    last_first = sharded.trajectories_done
    fence()
    t0 = time.perf_counter()
    sharded.run_minibatch(SEED, batch)
    fence()
    dt = max_over_ranks(dt_a + time.perf_counter() - t0)
    out["mccfr"] = {"metric": "ES-MCCFR trajectories/sec", "value": batch * nb / dt, "unit": "trajectories/s",
                    "seconds": dt, "scaling": "strong", "kernel": solver.last_kernel(),
                    "config": {"workload": f"leduc_poker external-sampling MCCFR, 2^24 trajectories in {nb} mini-batches "
                                           f"of 2^20 sharded over {world} rank(s), "
                                           + ("one all-reduce of 2 x [936,3] fp64 per mini-batch" if world > 1
                                              else "no collective at 1 GPU")}}
    # NashConv of the average policy after the 2^24 trajectories (device judge,
    # the same numbers as the oracle's TabularBestResponse to 1e-12: tests/test_gpu_cfr.py)
    out["mccfr"]["nash_conv_after"] = float(solver.nash_conv())
    if rank == 0:
        rf = mccfr_flat_roofline(batch * nb / dt / world, torch.cuda.get_device_properties(0).multi_processor_count * 4)
        if rf:
            out["mccfr"]["roofline"] = rf
    if snap is not None:
        try:
            impl, kind = cpu_checker()
            import parity
            t = solver.tables()
            reg0, cum0 = snap[0].cpu().numpy(), snap[1].cpu().numpy()
            scale = np.maximum(np.maximum(np.abs(reg0), np.abs(t["regrets"])), np.maximum(np.abs(cum0), np.abs(t["cum_policy"])))
            t0 = time.perf_counter()
            rec = parity.mccfr_minibatch(impl, "leduc_poker", t["keys"], t["nact"], reg0, t["regrets"] - reg0,
                                         t["cum_policy"] - cum0, SEED, last_first, batch, host_threads(), table_scale=scale)
            rec["cpu_seconds"] = time.perf_counter() - t0
            rec["against"] = ("UpdateRegrets (external_sampling_mccfr.cc:122-186) replayed on the frozen table over "
                              + ("oracle/_ref's game rules, SampleAction and regret matching" if kind == "reference"
                                 else "the oracle restatement") + f", {host_threads()} host threads")
            rec["what"] = (f"the last timed mini-batch (trajectories [{last_first}, {last_first + batch}), all ranks' shards): every "
                           "regret and average-policy cell of table-after minus table-before, tolerance 1e-11 of the cell's "
                           "|increment| mass + 8 ulps of the tables (summation order only)")
            out["mccfr"]["parity"] = rec
            out["mccfr"]["parity_checked_trajectories"] = rec["trajectories"]
        except AssertionError as e:
            out["mccfr"]["parity"] = {"error": str(e)}
            out["mccfr"]["parity_checked_trajectories"] = 0
        del snap
    if world > 1:  # the exchange step on its own: one all-reduce of the two delta tables
        flat = solver.mccfr_delta_flat()
        osd.allreduce_sum_(flat)
        fence()
        t0 = time.perf_counter()
        for _ in range(50):
            osd.allreduce_sum_(flat)
        fence()
        out["mccfr"]["allreduce_us"] = max_over_ranks(time.perf_counter() - t0) / 50 * 1e6
        out["mccfr"]["allreduce_bytes"] = int(flat.numel() * flat.element_size())
        out["mccfr"]["allreduce_backend"] = "torch.distributed " + dist.get_backend()
        # the same message through the one-shot all-reduce (peer-mapped windows, one launch, sums in rank order:
        # osg_comm_oneshot_*), and the 16 mini-batches again with it as the exchange step; a failure here is
        # reported, it never costs the line (the collective has a wall-clock timeout: no hang)
        try:
            one = osd.OneShotComm(ctx, flat.numel())
            for _ in range(10):
                one.allreduce_sum_(flat)
            fence()
            t0 = time.perf_counter()
            for _ in range(50):
                one.allreduce_sum_(flat)
            fence()
            shot = {"allreduce_us": max_over_ranks(time.perf_counter() - t0) / 50 * 1e6,
                    "what": "every rank pushes its 44 928 B into every peer's window over xGMI (hipIpc-mapped), waits for "
                            "the peers' chunks and sums the slots in rank order: one launch, bit-identical sums on all ranks"}
            s2 = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
            sh2 = osd.ShardedMccfr(s2, comm=one)
            sh2.run_minibatch(SEED, 1 << 12)
            fence()
            t0 = time.perf_counter()
            for _ in range(nb):
                sh2.run_minibatch(SEED, batch)
            fence()
            dt2 = max_over_ranks(time.perf_counter() - t0)
            shot["trajectories_per_s"] = batch * nb / dt2
            shot["nash_conv_after"] = float(s2.nash_conv())
            del sh2, s2
            one.close()
            out["mccfr"]["oneshot"] = shot
        except Exception as e:  # noqa: BLE001
            out["mccfr"]["oneshot"] = {"error": f"{type(e).__name__}: {e}"}
        del flat
    if rank == 0:
        t = solver.tables()
        out["mccfr"]["tables_finite"] = bool((abs(t["regrets"]) < 1e300).all())
    del solver, sharded
    # Quality per second (the throughput above says nothing about sample efficiency: a mini-batch shares one
    # frozen table).  Time — traversal + fold launches only, NashConv evaluations not timed — until the average
    # policy's NashConv drops below each threshold, at the mini-batch size the sweep in
    # profiles/r02_mccfr_quality.log found best (2^14), with the table refreshed after every mini-batch.
    out["mccfr"]["quality"] = mccfr_quality(osa, torch, dist, ctx, rank, world, 1 << 14, 1.5)
    # ---- the two callers either side of the path that SURVEY.md 8(f) names: the batched RL environment step and the
    #      policy judge (NashConv / exploitability of a tabular policy) ----
    if rank == 0:
        try:
            from open_spiel_amd._abi import check, lib
            n_env = 1 << 20
            eb = osa.StateBatch(ctx, "connect_four", n_env)
            should_reset = torch.ones(n_env, dtype=torch.uint8, device="cuda")
            cur = torch.empty(n_env, dtype=torch.int8, device="cuda")
            typ = torch.empty(n_env, dtype=torch.uint8, device="cuda")
            rew = torch.empty((n_env, 2), dtype=torch.float64, device="cuda")
            msk = torch.empty((n_env, 1), dtype=torch.int32, device="cuda")
            acts = torch.full((n_env,), -1, dtype=torch.int32, device="cuda")

            # the agent: the lowest legal column, as a 128-entry table lookup on the 7-bit mask (two torch launches — an index
            # cast and a gather — instead of the seven elementwise ones of round 4, which were 70 % of this loop's time; the
            # kernel under test is osg_env_step)
            lut = torch.tensor([-1] + [(m & -m).bit_length() - 1 for m in range(1, 128)], dtype=torch.int32, device="cuda")
            ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(200)]
            ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(200)]

            def env_step(t, k=None):
                if k is not None:
                    ev_a[k].record()
                check(lib().osg_env_step(eb._h, acts.data_ptr(), should_reset.data_ptr(), SEED, 0, t, cur.data_ptr(),
                                         typ.data_ptr(), rew.data_ptr(), msk.data_ptr()))
                if k is not None:
                    ev_b[k].record()
                torch.index_select(lut, 0, msk[:, 0].to(torch.int64), out=acts)
            for t in range(5):
                env_step(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 200
            for t in range(5, 5 + steps):
                env_step(t, t - 5)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            kern_us = sum(a.elapsed_time(b) for a, b in zip(ev_a, ev_b)) / steps * 1e3
            # the same launch back to back (an event pair around ONE ~11 us launch carries ~2 us of its own): 100 launches with
            # "no action" for every environment — the time step is formed and written again, the same 60 bytes move and the
            # step runs the same code — between one pair of events
            noop = torch.full((n_env,), -1, dtype=torch.int32, device="cuda")
            keep = torch.zeros(n_env, dtype=torch.uint8, device="cuda")
            for _ in range(5):
                check(lib().osg_env_step(eb._h, noop.data_ptr(), keep.data_ptr(), SEED, 0, 0, cur.data_ptr(), typ.data_ptr(),
                                         rew.data_ptr(), msk.data_ptr()))
                keep.zero_()
            b2b_a, b2b_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            b2b_a.record()
            for _ in range(100):
                check(lib().osg_env_step(eb._h, noop.data_ptr(), keep.data_ptr(), SEED, 0, 0, cur.data_ptr(), typ.data_ptr(),
                                         rew.data_ptr(), msk.data_ptr()))
            b2b_b.record()
            torch.cuda.synchronize()
            b2b_us = b2b_a.elapsed_time(b2b_b) / 100 * 1e3
            # the compact side arrays (osg_env_step_compact, round 6: u8 actions, one in/out flag byte, i8 rewards holding twice
            # the return): 41 bytes per environment and step, measured the same way on the same batch
            c_act = torch.full((n_env,), 255, dtype=torch.uint8, device="cuda")
            c_flag = torch.ones(n_env, dtype=torch.uint8, device="cuda")
            c_rew = torch.empty((n_env, 2), dtype=torch.int8, device="cuda")
            for _ in range(5):
                check(lib().osg_env_step_compact(eb._h, c_act.data_ptr(), c_flag.data_ptr(), SEED, 0, 0, c_rew.data_ptr(), msk.data_ptr()))
            torch.cuda.synchronize()
            b2b_a.record()
            for _ in range(100):
                check(lib().osg_env_step_compact(eb._h, c_act.data_ptr(), c_flag.data_ptr(), SEED, 0, 0, c_rew.data_ptr(), msk.data_ptr()))
            b2b_b.record()
            torch.cuda.synchronize()
            compact_us = b2b_a.elapsed_time(b2b_b) / 100 * 1e3
            compact_bytes = 16 + 16 + 1 + 1 + 1 + 2 + 4
            # bytes one connect_four environment moves per step: state in + out (2 x 16), action 4, should_reset 1 + 1,
            # current player 1, step type 1, rewards 2 x 8, mask word 4
            env_bytes = 16 + 16 + 4 + 2 + 1 + 1 + 16 + 4
            out["env_step"] = {"metric": "RL environment steps/sec (osg_env_step: reset / apply / chance / time step fields)",
                               "value": n_env * steps / dt, "unit": "env-steps/s", "us_per_launch": dt / steps * 1e6,
                               "workload": f"connect_four, {n_env} environments, {steps} synchronous steps of every environment with a "
                                           "table-lookup torch agent in between (python/rl_environment.py:379-418 per environment in the reference)",
                               "roofline": {"kernel": "k_env_step_x2", "bound": "infinity_cache (60 B x 2^20 = 63 MB per launch stays on the chip; peak = the HBM figure)",
                                            "algorithmic_bytes_per_env_step": env_bytes, "kernel_us_per_launch": kern_us,
                                            "achieved": env_bytes * n_env / (kern_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": env_bytes * n_env / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                            "kernel_share_of_step": kern_us / (dt / steps * 1e6),
                                            "kernel_us_back_to_back": b2b_us,
                                            "frac_back_to_back": env_bytes * n_env / (b2b_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                            "compact": {"kernel": "k_env_step_compact_x2", "bytes_per_env_step": compact_bytes,
                                                        "kernel_us_back_to_back": compact_us,
                                                        "env_steps_per_s": n_env / (compact_us * 1e-6),
                                                        "frac_back_to_back": compact_bytes * n_env / (compact_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
                                            "note": "frac: HIP events around the osg_env_step launch of every timed step (one pair per ~11 us launch: "
                                                    "~2 us of the pair itself inside); frac_back_to_back: 100 launches between one pair of events, "
                                                    "every environment stepped with 'no action' (the same bytes, the same straight-line step). "
                                                    "Rewards leave as float64 [n, P] and actions arrive as int32 (the reference's TimeStep types): "
                                                    "20 of the 60 bytes.  (The same step on the headline kernel's fused connect_four step measured no "
                                                    "faster — 164.8 vs 165.0 us per 2^24 environments: the launch is bound by its bytes — and was dropped: "
                                                    "profiles/r05p_env_step_fused_ab.txt)"}}
            del eb, lut, ev_a, ev_b, noop, keep, c_act, c_flag, c_rew
            judge = {}
            for g in ("kuhn_poker", "leduc_poker"):
                sj = osa.TabularSolver(ctx, g)
                sj.evaluate_and_update_policy(20)
                sj.nash_conv()
                ctx.synchronize()
                t0 = time.perf_counter()
                reps = 50
                for _ in range(reps):
                    sj.nash_conv()
                dt = time.perf_counter() - t0
                judge[g] = {"us_per_nash_conv": dt / reps * 1e6, "nash_conv_after_20_iterations": sj.nash_conv()}
                # CFRBRSolver (cfr_br.cc:48-83) on the same tree: best responses of every player, then one pass per player
                sb = osa.TabularSolver(ctx, g)
                sb.evaluate_and_update_policy_cfr_br(5)
                ctx.synchronize()
                t0 = time.perf_counter()
                sb.evaluate_and_update_policy_cfr_br(300)
                ctx.synchronize()
                judge[g]["cfr_br_iterations_per_s"] = 300 / (time.perf_counter() - t0)
                judge[g]["cfr_br_nash_conv_after_305"] = sb.nash_conv()
                del sj, sb
            try:   # the largest tree served: a launch per level and phase (k_geval_*; one workgroup walked it in 33 ms)
                s3 = osa.TabularSolver(ctx, "leduc_poker(players=3)")
                s3.evaluate_and_update_policy(2)
                s3.nash_conv()
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    nc3 = s3.nash_conv()
                judge["leduc_poker(players=3)"] = {"ms_per_nash_conv": (time.perf_counter() - t0) / 5 * 1e3,
                                                   "nash_conv_after_2_iterations": nc3, "histories": 1831601}
                del s3
            except Exception as e:  # noqa: BLE001
                judge["leduc_poker(players=3)"] = {"error": f"{type(e).__name__}: {e}"}
            out["policy_evaluation"] = {"metric": "NashConv evaluations (osg_cfr_evaluate_policy: expected returns + one best response "
                                                  "per player on the flattened tree)", "per_game": judge,
                                        "note": "host call to host result; the tables stay on the device (the average policy is "
                                                "formed inside the kernel). leduc_poker: k_eval_jobs, 30 expected-returns jobs + 12 "
                                                "best-response jobs of one workgroup each, the last one sums the chance levels (222 us per "
                                                "call on one workgroup before); leduc_poker(players=3), 1.83 M histories: k_geval_* — one "
                                                "bottom-up sweep for every player's best response and the expected returns (39 launches), the "
                                                "argmax by a wavefront per infostate (2.0 ms per call in round 4); the reference walks the "
                                                "tree per call (tabular_exploitability.cc:30-89)"}
        except Exception as e:  # noqa: BLE001
            out["env_step"] = {"error": f"{type(e).__name__}: {e}"}
        # ---- the byte-bound step of the largest record served: hex(9), SURVEY.md 8(d)'s 109 B per state-step ----
        # 2^24 states (805 MB of records in + 805 MB out per launch: every byte through HBM) is the figure; the 2^22-state
        # launch (403 MB, 1.5 x the Infinity Cache) is kept beside it, labelled as partly cache-resident.
        try:
            hex_legs = {}
            for n_hex in (1 << 24, 1 << 22):
                hb = osa.StateBatch(ctx, "hex(board_size=9)", n_hex)
                hacts, _depth = hb.synth(SEED, 60)            # state i: hash(i) mod 60 random legal moves, one legal action
                hd = osa.StateBatch(ctx, "hex(board_size=9)", n_hex)
                hstatus = torch.empty(n_hex, dtype=torch.uint8, device="cuda")
                launches = 20 if n_hex > (1 << 22) else 50
                for _ in range(3):
                    hb.step(hacts, dst=hd, status=hstatus, want_mask=False)
                h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                h0.record()
                for _ in range(launches):
                    hb.step(hacts, dst=hd, status=hstatus, want_mask=False)
                h1.record()
                torch.cuda.synchronize()
                assert int((hstatus & 0x40).sum().item()) == 0, "synthetic hex actions must all be legal"
                hus = h0.elapsed_time(h1) / launches * 1e3
                rec_bytes = hb.desc.state_words * 4
                moved = 2 * rec_bytes + 2                     # record in, record out, action, status
                hex_parity = None
                if with_cpu and n_hex == (1 << 24):
                    try:
                        hex_parity = parity_check(torch, hb, hd, hacts, None, hstatus, SEED, 0, 1 << 16,
                                                  game="hex(board_size=9)", depth_mod=60)
                    except AssertionError as e:
                        hex_parity = {"error": str(e)[:500], "states": 0}
                hex_legs[n_hex] = {"parity": hex_parity, "states": n_hex, "kernel_us_per_launch": hus, "launches": launches, "record_bytes": rec_bytes,
                                   "value": n_hex / hus * 1e6, "bytes_moved_per_state_step": moved,
                                   "frac_on_bytes_moved": moved * n_hex / hus / 1e3 / HBM_PEAK_GBS,
                                   "frac_on_survey_bytes": 109 * n_hex / hus / 1e3 / HBM_PEAK_GBS}
                del hb, hd, hacts, hstatus
            big, small = hex_legs[1 << 24], hex_legs[1 << 22]
            out["hex_step"] = {"metric": "hex(board_size=9) fused legality + ApplyAction + status, states/sec",
                               "value": big["value"], "unit": "state-steps/s", "states": big["states"],
                               "kernel_us_per_launch": big["kernel_us_per_launch"], "record_bytes": big["record_bytes"],
                               "parity": big["parity"], "parity_checked_states": (big["parity"] or {}).get("states", 0),
                               "roofline": {"bound": "hbm", "kernel": "k_step_hexvec",
                                            "bytes_moved_per_state_step": big["bytes_moved_per_state_step"],
                                            "achieved": big["bytes_moved_per_state_step"] * big["states"] / big["kernel_us_per_launch"] / 1e3,
                                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac_on_bytes_moved": big["frac_on_bytes_moved"],
                                            "algorithmic_bytes_per_state_step": 109, "frac": big["frac_on_survey_bytes"],
                                            "note": "frac_on_bytes_moved is the traffic-true fraction (the 12-word record moves 98 B per "
                                                    "state-step; SURVEY.md 8(d) prices a 13-word record: 109 B, `frac`); launches back to "
                                                    "back between one event pair; the successor's mask row is not written (on a hex board "
                                                    "it is ~occupied of the successor record)"},
                               "infinity_cache_partial": dict(small, note="403 MB per launch against a 256 MiB Infinity Cache: part of the "
                                                                         "traffic never reaches HBM; not an HBM figure")}
        except Exception as e:  # noqa: BLE001
            out["hex_step"] = {"error": f"{type(e).__name__}: {e}"}
    # ---- config 1: tic_tac_toe MCTSBot(RandomRolloutEvaluator(20, 42), 1000 sims, solve) — plumbing ----
    if rank == 0:
        out["ttt_mcts"] = ttt_mcts_config1(with_cpu)
    if with_cpu and rank == 0:
        impl, kind = cpu_checker()
        threads = host_threads()
        g = impl.Game("hex(board_size=9)")
        per_thread = 64                                                             # ~10 s of CPU work
        secs, done_cpu = g.bench_mcts(SEED, threads * per_thread, 40, 1024, 1, 2.0, threads)
        out["mcts"]["cpu_baseline"] = {"value": done_cpu / secs, "unit": "sims/s", "cores": threads, "kind": kind,
                                       "sample": f"{threads * per_thread} roots x 1024 sims, one MCTSBot per root, "
                                                 f"{threads} threads, {secs:.1f} s"}
        gk = impl.Game("kuhn_poker")
        secs = gk.bench_cfr(0, 100000, 1)
        out["cfr"]["cpu_baseline"] = {"value": 100000 / secs, "unit": "iterations/s", "cores": 1, "kind": kind,
                                      "sample": f"100000 CFRSolver iterations, 1 thread, {secs:.2f} s"}
        # the replicas beside the same number of independent CPU solvers as there are cores
        secs_r = gk.bench_cfr(0, 50000, threads)
        out["cfr"]["replicas"]["cpu_baseline"] = {
            "value": threads * 50000 / secs_r, "unit": "solver-iterations/s", "cores": threads, "kind": kind,
            "sample": f"{threads} independent CFRSolver objects, one per thread, 50000 iterations each, {secs_r:.2f} s"}
        gl = impl.Game("leduc_poker")
        if isinstance(out["cfr"].get("leduc"), dict) and "value" in out["cfr"]["leduc"]:
            secs = gl.bench_cfr(0, 150, 1)
            out["cfr"]["leduc"]["cpu_baseline"] = {"value": 150 / secs, "unit": "iterations/s", "cores": 1, "kind": kind,
                                                   "sample": f"150 CFRSolver iterations on leduc_poker, 1 thread, {secs:.2f} s"}
        secs = gl.bench_cfr(2, 100000, 1)
        out["mccfr"]["cpu_baseline"] = {"value": 200000 / secs, "unit": "trajectories/s", "cores": 1, "kind": kind,
                                        "sample": f"100000 RunIteration (= 200000 traversals), 1 thread, {secs:.2f} s"}
        # the same solver by quality: seconds of RunIteration until NashConv <= threshold (about 5 s of CPU work)
        sol = impl.Solver(gl, "mccfr_simple", SEED)
        reached, spent, it, step = {}, 0.0, 0, 200
        while spent < 8.0 and len(reached) < len(NASH_CONV_THRESHOLDS):
            t0 = time.perf_counter()
            sol.iterate(step)
            spent += time.perf_counter() - t0
            it += step
            nc = sol.nash_conv()
            for th in NASH_CONV_THRESHOLDS:
                if nc <= th and str(th) not in reached:
                    reached[str(th)] = {"seconds": spent, "iterations": it}
            step = max(200, it // 4)
        out["mccfr"]["cpu_baseline"]["quality"] = {
            "seconds_to_nash_conv": reached, "iterations": it, "seconds": spent, "nash_conv": sol.nash_conv(),
            "note": "ExternalSamplingMCCFRSolver::RunIteration is sequential by construction: 1 thread"}
        q = out["mccfr"].get("quality")
        if q:
            q["speedup_vs_cpu_at_equal_nash_conv"] = {
                th: reached[th]["seconds"] / q["seconds_to_nash_conv"][th]["seconds"]
                for th in reached if th in q["seconds_to_nash_conv"]}
    host_barrier()
    return out


NASH_CONV_THRESHOLDS = (1.0, 0.3, 0.1)


def source_is_current(profile_relpath):
    """True / False: does the committed profile a roofline quotes still describe the kernel as it is built now
    (tools/profile_sources.py: sha256 of the defining sources recorded beside the profile)?  None: not a stamped kind."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import profile_sources
        return profile_sources.is_current(profile_relpath)
    except Exception:  # noqa: BLE001
        return None


def solver_counters():
    """Per-launch counters of the kernels of configs 3 and 5 from the newest committed profile
    (profiles/r*_pmc_solvers.json, written by tools/pmc_solvers.sh on the GPU box), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_solvers.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            rec = json.load(f)
        rec["file"] = os.path.relpath(files[-1], ROOT)
        return rec
    except (OSError, ValueError):
        return None


# Issue intervals measured on this chip by tools/clock_probe.hip (profiles/r02_clock_probe.log): a dependent vector
# instruction of a LONE wavefront every 3.737 ns; with every SIMD saturated a simple vector instruction every 1.03 ns and
# a scalar-unit instruction (ALU or branch) every 1.833 ns per SIMD.
LONE_WAVE_ISSUE_NS, VALU_ISSUE_NS, SALU_ISSUE_NS = 3.737, 1.03, 1.833


def cfr_small_roofline(iterations_per_s):
    """Config 3's bound: k_cfr_small is ONE wavefront (58 histories <= 64 lanes) running a dependent chain, so its
    ceiling is a lone wavefront's instruction issue, not bytes (< 8 KB per iteration, all LDS)."""
    pc = solver_counters()
    if not pc or "k_cfr_small" not in pc:
        return None
    u = pc["k_cfr_small"]["per_unit"]
    classes = {k: u.get("SQ_INSTS_" + k, 0.0) for k in ("VALU", "SALU", "LDS", "SMEM", "VMEM_RD", "VMEM_WR", "BRANCH")}
    total = sum(classes.values())
    ns = 1e9 / iterations_per_s
    # one wavefront gets an issue slot every 4 cycles of its SIMD (2.4 GHz: 1.667 ns) when the instruction does not wait
    # for the one before it, and issues a DEPENDENT instruction every 3.737 ns (tools/clock_probe.hip): the kernel lies
    # between the two — frac is quoted against the first (the bound no schedule of these instructions can beat)
    slot_ns = 4.0 / 2.4
    return {"bound": "instruction issue of one wavefront (58 histories fit 64 lanes: the solver is one wavefront)",
            "instructions_per_iteration": total, "by_class": classes,
            "issue_slot_ns": slot_ns, "issue_ns_per_iteration_at_least": total * slot_ns,
            "dependent_issue_ns_per_instruction": LONE_WAVE_ISSUE_NS,
            "issue_ns_per_iteration_if_every_instruction_waited": total * LONE_WAVE_ISSUE_NS,
            "measured_ns_per_iteration": ns, "measured_ns_per_instruction": ns / total if total else None,
            "frac": total * slot_ns / ns,
            "lds_instructions_per_iteration": classes["LDS"],
            "wait_share_of_wave_cycles": (u.get("SQ_WAIT_ANY", 0.0) / u["SQ_WAVE_CYCLES"]) if u.get("SQ_WAVE_CYCLES") else None,
            "algorithmic_bytes_per_iteration": "< 8 KB, LDS-resident (SURVEY.md 8(d)): an HBM roofline does not apply",
            "source": pc["file"] + " (" + pc.get("source", "") + ")", "source_current": source_is_current(pc["file"]),
            "note": "the lever is the instruction count (round 5: 1 586 -> ~800 per iteration, 1.9e5 -> 3.6e5 it/s); counters "
                    "are those of the committed profile named in `source` — a kernel changed since then shows as "
                    "measured_ns_per_instruction outside 1.7-3.7 ns; the replicas figure is the same kernel with every "
                    "SIMD holding several wavefronts"}


def mccfr_flat_roofline(trajectories_per_s, simds):
    """Config 5's kernel (k_mccfr_resident_flat: one trajectory per lane, policy / delta tables in LDS, the read-only tree
    records through L2 so that two 1024-lane workgroups share a CU, the traversal's frames below the top two in a per-lane
    scratch stack): issue-rate fractions and where the waves wait."""
    pc = solver_counters()
    if not pc or "k_mccfr_resident_flat" not in pc:
        return None
    k = pc["k_mccfr_resident_flat"]
    c, n = k["counters_per_launch"], k["units_per_launch"]
    secs = n / trajectories_per_s
    valu_s = c.get("SQ_INSTS_VALU", 0.0) / simds * VALU_ISSUE_NS * 1e-9
    salu_s = (c.get("SQ_INSTS_SALU", 0.0) + c.get("SQ_INSTS_BRANCH", 0.0)) / simds * SALU_ISSUE_NS * 1e-9
    waves = c.get("SQ_WAVES", 0.0) or 1.0
    rec = {"trajectories_per_lane": n / (waves * 64.0), "waves_per_simd": waves / simds,
           "wave_instructions_per_wave": {x: c.get("SQ_INSTS_" + x, 0.0) / waves for x in ("VALU", "SALU", "BRANCH", "LDS", "VMEM_RD", "VMEM_WR", "SMEM")},
           "scratch_frame_instructions_per_wave": (c.get("SQ_INSTS_VMEM_RD", 0.0) + c.get("SQ_INSTS_VMEM_WR", 0.0)) / waves,
           "lds_instructions_per_wave": c.get("SQ_INSTS_LDS", 0.0) / waves,
           "vector_issue_seconds_at_least": valu_s, "scalar_issue_seconds_at_least": salu_s, "seconds_per_mini_batch": secs,
           "frac_of_vector_issue_bound": valu_s / secs, "frac_of_scalar_issue_bound": salu_s / secs,
           "wait_any_share_of_wave_cycles": (c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
           "wait_inst_share_of_wave_cycles": (c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
           "lds_bank_conflict_cycles_per_lds_active_cycle": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_ACTIVE_INST_LDS"]) if c.get("SQ_ACTIVE_INST_LDS") else None,
           "algorithmic_bytes_per_trajectory": "0.5-1 KB of table rows, served from LDS (SURVEY.md 8(d): HBM roofline N/A)",
           "source": pc["file"] + " (" + pc.get("source", "") + ")", "source_current": source_is_current(pc["file"])}
    top = max(("vector-unit instruction issue", rec["frac_of_vector_issue_bound"]), ("scalar-unit instruction issue", rec["frac_of_scalar_issue_bound"]),
              key=lambda t: t[1])
    wait = rec["wait_any_share_of_wave_cycles"] or 0.0
    rec["bound"] = (top[0] if top[1] >= 0.6 else
                    f"latency of the traversal's dependent chain (waves parked {wait:.0%} of their cycles; the busiest issue pipe is at {top[1]:.2f})")
    return rec


def mcts_instruction_mix():
    """SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_BRANCH per simulation of the hex(9) search kernel from the newest committed
    profiles/r*_pmc_k_mcts_wave_hex9_8192x1024.csv, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_k_mcts_wave_hex9_8192x1024.csv")))
    if not files:
        return None
    vals = {}
    with open(files[-1]) as f:
        for ln in f:
            parts = ln.split(",")
            if len(parts) >= 3 and parts[0] in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH"):
                vals[parts[0]] = float(parts[2])
    if "SQ_INSTS_VALU" not in vals or "SQ_INSTS_SALU" not in vals:
        return None
    return {"valu": vals["SQ_INSTS_VALU"], "salu": vals["SQ_INSTS_SALU"], "branch": vals.get("SQ_INSTS_BRANCH", 0.0),
            "source": os.path.relpath(files[-1], ROOT)}


def mccfr_quality(osa, torch, dist, ctx, rank, world, batch, budget_s):
    """leduc_poker ES-MCCFR by quality: seconds until the average policy's NashConv (device judge) is below each
    threshold, mini-batches of `batch` trajectories (global; sharded over the ranks, one all-reduce of the delta
    tables each), for the synchronous schedule (sample -> all-reduce -> fold) and the overlapped one
    (ShardedMccfr(overlap=True): mini-batch k + 1 is sampled while k's deltas are summed; tables stale by one
    mini-batch).  Time = the slowest rank's wall time around the mini-batches (synchronised on both sides), the
    NashConv evaluations are not timed."""
    from open_spiel_amd import distributed as osd

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run(overlap):
        s = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
        s.run_mccfr(SEED, 64)
        s.reset()
        sharded = osd.ShardedMccfr(s, overlap=overlap)
        if overlap:   # allocate the two delta buffers outside the timed region
            sharded.run_minibatch(SEED, 64)
            sharded.finish()
            s.reset()
            sharded.trajectories_done = 0
        reached, spent, updates, check = {}, 0.0, 0, 1
        while spent < budget_s and len(reached) < len(NASH_CONV_THRESHOLDS):
            todo = check - updates
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(todo):
                sharded.run_minibatch(SEED, batch)
            sharded.finish()
            torch.cuda.synchronize()
            spent += max_over_ranks(time.perf_counter() - t0)
            updates = check
            nc = s.nash_conv()
            for th in NASH_CONV_THRESHOLDS:
                if nc <= th and str(th) not in reached:
                    reached[str(th)] = {"seconds": spent, "mini_batches": updates, "trajectories": updates * batch}
            check = max(check + 1, int(check * 1.25))
        return {"seconds_to_nash_conv": reached, "mini_batches": updates, "seconds": spent, "nash_conv": s.nash_conv(),
                "us_per_mini_batch": spent / max(updates, 1) * 1e6}

    sync = run(False)
    over = run(True)
    out = dict(sync)
    out.update({"mini_batch": batch, "world": world,
                "schedule": f"every mini-batch: 2^{batch.bit_length() - 1} traversals sharded over {world} rank(s) against the "
                            "frozen table, one all-reduce of 2 x [936, 3] fp64 (none at 1 rank), one fold; thresholds "
                            "checked at geometrically spaced mini-batch counts (x1.25), the evaluations are not timed",
                "overlapped": dict(over, schedule="two delta buffers: mini-batch k + 1 is sampled while mini-batch k's deltas "
                                                  "are all-reduced on the collective's own stream and folded on arrival "
                                                  "(tables stale by one mini-batch; pending deltas folded before each "
                                                  "evaluation)" + ("" if world > 1 else
                                                                   "; at 1 rank there is no collective to hide: this row shows what the staleness costs")),
                })
    return out


def ttt_mcts_config1(with_cpu):
    """BASELINE.json configs[0] (plumbing): tic_tac_toe MCTSBot(RandomRolloutEvaluator(20, 42), uct_c = 2, 1000
    simulations, max_memory_mb = 5, solve, seed 42) — the mcts_test.cc:35-49 setup — from the initial state and the
    three MCTS-Solver positions of mcts_test.cc:126-155: the CPU reference's simulations/s and, beside it, what ONE
    such search costs through the device path's single-root MCTSBot::Step (the drop-in class of the host mirror;
    a single search is a latency case for a GPU — the batch entry points are the product)."""
    out = {"config": "tic_tac_toe MCTSBot(RandomRolloutEvaluator(20, 42), uct_c=2, 1000 sims, max_memory_mb=5, solve, seed 42) "
                     "from the initial state and the 3 solver positions of mcts_test.cc:126-155"}
    positions = ("", "x(1,1) o(0,0) x(2,2)", "x(1,1) o(0,0) x(2,2) o(0,1) x(0,2)", "x(0,1) o(2,2)")
    try:
        from open_spiel_amd import pyspiel_hip as ps
        game = ps.load_game("tic_tac_toe")
        per = []
        for pos in positions:
            state = game.new_initial_state()
            for tok in pos.split():
                state.apply_action(next(a for a in state.legal_actions()
                                        if state.action_to_string(state.current_player(), a) == tok))
            bot = ps.MCTSBot(game, ps.RandomRolloutEvaluator(20, 42), 2.0, 1000, 5, True, 42, False)
            bot.step(state)                               # warm-up (pool allocation)
            t0 = time.perf_counter()
            reps = 3
            sims = 0
            for _ in range(reps):
                root = bot.mcts_search(state)
                sims += root.explore_count
            dt = time.perf_counter() - t0
            per.append({"position": pos or "initial", "us_per_search": dt / reps * 1e6, "simulations_per_search": sims / reps})
        tot_s = sum(p["us_per_search"] for p in per) * 1e-6
        tot_sims = sum(p["simulations_per_search"] for p in per)
        out["device_single_root"] = {"value": tot_sims / tot_s, "unit": "sims/s", "per_position": per,
                                     "what": "pyspiel_hip.MCTSBot.mcts_search on ONE root (osg_mcts_tree_*: the whole search is one launch; the "
                                             "first 6144 nodes of the tree live in LDS, the searching wavefront runs in lockstep — a node's "
                                             "children are valued one per lane, its 20 playouts played one per lane), whole SearchNode tree "
                                             "downloaded; a lone wavefront retires an instruction every ~8 cycles, so a strictly sequential "
                                             "1000-simulation search stays behind one host core: the batch entry points are the product"}
    except Exception as e:  # noqa: BLE001
        out["device_single_root"] = {"error": f"{type(e).__name__}: {e}"}
    if with_cpu:
        try:
            impl, kind = cpu_checker()
            secs, sims = impl.Game("tic_tac_toe").bench_mcts_config1(1000, 40)
            out["cpu_baseline"] = {"value": sims / secs, "unit": "sims/s", "cores": 1, "kind": kind,
                                   "sample": f"40 x 4 MCTSearch calls (initial state + the 3 solver positions), {sims} simulations, "
                                             f"1 thread, {secs:.2f} s"}
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def committed_random_steps_mix():
    """Vector instructions per env step of the K = 32 random-steps kernel from the newest committed counter
    profile (profiles/r*_pmc_k_random_steps.json, written by tools/gpu_validation.sh), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_k_random_steps.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        rec = json.load(f)
    rec["source"] = os.path.relpath(files[-1], ROOT)
    return rec if "valu_per_env_step" in rec else None


def pmc_traffic():
    """HBM bytes per launch of the headline kernel from the committed PMC profile, if any."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None, None
    with open(path) as f:
        rec = json.load(f)
    return rec.get("bytes_per_launch"), rec.get("source"), (rec.get("dram_leg") or {}).get("bytes_per_launch")


def timed_launches(torch, launch, launches, warmup):
    """Average duration of one launch: HIP events on the launching stream (the context is bound to
    torch's current stream) around `launches` back-to-back launches, after `warmup` untimed ones."""
    for _ in range(warmup):
        launch()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(launches):
        launch()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / 1e3 / launches


def copy_ceiling(osa, torch, ctx, n, launches, warmup):
    """16-byte-per-lane copy (non-temporal stores) of the bytes one launch over n states moves (35 B per state: half read,
    half written) on the same stream: seconds per copy."""
    from open_spiel_amd._abi import check, lib
    half = (ALGO_BYTES_PER_STEP * n // 2) // 16 * 16
    a = torch.empty(half, dtype=torch.uint8, device="cuda")
    b = torch.empty(half, dtype=torch.uint8, device="cuda")
    a.zero_()
    secs = timed_launches(torch, lambda: check(lib().osg_copy_bytes(ctx._h, b.data_ptr(), a.data_ptr(), half)),
                          launches, warmup)
    return secs, 2 * half


def dram_leg(osa, torch, ctx, src, actions):
    """The fused step over 2^24 states (the 2^20 positions tiled 16 times): 587 MB per launch, far beyond
    the 256 MiB Infinity Cache, so every byte comes from / goes to HBM."""
    n, big = src.n, DRAM_LEG_STATES
    idx = torch.arange(big, device="cuda", dtype=torch.int64) % n
    src_big = src.gather(idx)
    act_big = actions[idx].contiguous()
    del idx
    dst_big = osa.StateBatch(ctx, "connect_four", big)
    mask, status = src_big.step_buffers()
    secs = timed_launches(torch, lambda: src_big.step(act_big, dst=dst_big, mask=mask, status=status), 100, 20)
    assert int((status & 0x40).sum().item()) == 0
    del src_big, dst_big, mask, status, act_big
    csecs, cbytes = copy_ceiling(osa, torch, ctx, big, 100, 20)
    achieved = ALGO_BYTES_PER_STEP * big / secs / 1e9
    copy_gbs = cbytes / csecs / 1e9
    return {"states": big, "bound": "hbm", "algorithmic_bytes_per_launch": ALGO_BYTES_PER_STEP * big,
            "avg_launch_us": secs * 1e6, "launches": 100, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "copy_ceiling": {"gbs": copy_gbs, "us": csecs * 1e6, "bytes": cbytes,
                             "what": "osg_copy_bytes: uint4 copy (non-temporal stores) of the same number of bytes, same stream"},
            "frac_of_copy_ceiling": achieved / copy_gbs,
            "copy_note": "the copy is a measured kernel (16 B per lane, workgroups of 256, non-temporal stores), not a "
                         "bound: where the fraction passes 1 the step kernel (workgroups of 128) simply outruns it"}


def persistent_leg(osa, torch, ctx, src, rank):
    """SURVEY.md 8(d) item 2, second figure: K = 32 uniformly random env steps per launch with the state
    in registers (osg_random_steps: on-device sampling, auto-reset of finished games)."""
    b = src.clone()
    counters = torch.zeros(2, dtype=torch.int64, device="cuda")
    k, launches = 32, 20
    for _ in range(3):
        b.random_steps(SEED, k, counters, index_offset=rank * src.n)
    torch.cuda.synchronize()
    before = int(counters[0].item())
    t0 = time.perf_counter()
    for j in range(launches):
        b.random_steps(SEED + 1 + j, k, counters, index_offset=rank * src.n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = int(counters[0].item()) - before
    return {"metric": "env-steps/s, K random steps per launch (state in registers, on-device sampling, auto-reset)",
            "value": steps / dt, "unit": "env-steps/s", "k": k, "launches": launches, "seconds": dt,
            "env_steps": steps, "us_per_launch": dt / launches * 1e6,
            "hbm_bytes_per_env_step": 32.0 / k,
            "note": "per rank; state read and written once per launch (32 B / K per env step), so this leg is "
                    "vector-issue bound, not memory bound"}


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn_ranks(gpus):
    """`python bench.py --gpus N` started as ONE process: become N ranks (one per GPU) by running this same file
    under torch.distributed.run with the same arguments — the command the module docstring shows, on a free
    127.0.0.1 port — and pass its exit code on.  Rank 0 of that run prints the line."""
    import subprocess
    env = dict(os.environ, OSG_BENCH_SPAWNED="1", MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


PMC_KERNELS = ("k_step_c4std", "k_random_steps")


def pmc_child(torch, osa):
    """The launches the in-run counter passes look at, and nothing else: 12 of the headline kernel over 2^20
    states, 12 over 2^24, 4 of the K = 32 random-steps kernel (bench.py --pmc-child, run under rocprofv3 --pmc)."""
    ctx = osa.Context(0)
    src, actions = synth_batch(osa, torch, ctx, STATES_PER_GPU, SEED, 0)
    dst = osa.StateBatch(ctx, "connect_four", src.n)
    mask, status = src.step_buffers()
    for _ in range(12):
        src.step(actions, dst=dst, mask=mask, status=status)
    b = src.clone()
    counters = torch.zeros(2, dtype=torch.int64, device="cuda")
    for j in range(4):
        b.random_steps(SEED + j, 32, counters, index_offset=0)
    torch.cuda.synchronize()
    print(json.dumps({"random_env_steps": int(counters[0].item()), "random_launches": 4}), flush=True)
    del dst, mask, status, b
    free_b, _total = torch.cuda.mem_get_info()
    if free_b > 4 * ALGO_BYTES_PER_STEP * DRAM_LEG_STATES:
        idx = torch.arange(DRAM_LEG_STATES, device="cuda", dtype=torch.int64) % src.n
        src_big, act_big = src.gather(idx), actions[idx].contiguous()
        del idx
        dst_big = osa.StateBatch(ctx, "connect_four", DRAM_LEG_STATES)
        mask, status = src_big.step_buffers()
        for _ in range(12):
            src_big.step(act_big, dst=dst_big, mask=mask, status=status)
    torch.cuda.synchronize()


def measure_pmc(timeout_s=150):
    """HBM traffic of the headline kernel and the instruction mix of the random-steps kernel, measured in THIS run:
    separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_INSTS_VALU + SQ_INSTS_SALU — FETCH_SIZE and WRITE_SIZE
    do not fit one pass, MI355X_MICROARCH.md) over `bench.py --pmc-child`, --kernel-trace only beside the
    counters.  FETCH_SIZE is doubled as the guide prescribes for gfx950 (128-byte requests of coalesced reads
    tallied at 64 B; checked in round 1 against a copy of known size), WRITE_SIZE taken as is.  Returns a dict
    or None (no rocprofv3, already under a profiler, a pass failed: the caller falls back to the committed file)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if os.environ.get("OSG_BENCH_NO_PMC") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    tmp = tempfile.mkdtemp(prefix="osg_pmc_", dir="/tmp")
    got, child_note = {}, {}
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_INSTS_SALU")):
            out = os.path.join(tmp, counters[0])
            cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child"]
            env = dict(os.environ, TMPDIR="/tmp", OSG_BENCH_NO_PMC="1")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID"):
                env.pop(k, None)
            p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd="/tmp",
                                 start_new_session=True)
            try:
                stdout, _ = p.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)     # the process group this call started, nothing else
                p.wait()
                return None
            if p.returncode != 0:
                return None
            for ln in stdout.splitlines():
                if ln.startswith("{"):
                    child_note = json.loads(ln)
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"]
                    kern = next((k for k in PMC_KERNELS if k in name), None)
                    if kern is None or r["Counter_Name"] not in counters:
                        continue
                    states = int(r["Grid_Size"]) * (2 if "k_step_c4std2" in name else 1)
                    got.setdefault((kern, states, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    except Exception as e:  # noqa: BLE001 - a counter pass must never cost the line
        print(f"[bench] in-run counter passes failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def mean(kern, states, counter):
        vals = got.get((kern, states, counter))
        if not vals:
            return None
        vals = vals[len(vals) // 4:]           # drop the first launches (cold caches)
        return sum(vals) / len(vals)

    res = {"source": "rocprofv3 --pmc passes run by this bench.py over `bench.py --pmc-child` (FETCH_SIZE, WRITE_SIZE, "
                     "SQ_INSTS_VALU + SQ_INSTS_SALU: three separate passes, --kernel-trace only beside them); FETCH_SIZE "
                     "(KB) doubled per MI355X_MICROARCH.md, WRITE_SIZE (KB) as is"}
    for key, states in (("step", STATES_PER_GPU), ("dram", DRAM_LEG_STATES)):
        f, w = mean("k_step_c4std", states, "FETCH_SIZE"), mean("k_step_c4std", states, "WRITE_SIZE")
        if f is not None and w is not None:
            res[key] = {"bytes_per_launch": f * 1024 * 2 + w * 1024, "fetch_bytes": f * 2048, "write_bytes": w * 1024,
                        "ratio_to_algorithmic": (f * 2048 + w * 1024) / (ALGO_BYTES_PER_STEP * states)}
    rs_keys = [k for k in got if k[0] == "k_random_steps" and k[2] == "SQ_INSTS_VALU"]
    if rs_keys and child_note.get("random_env_steps"):
        kern, grid, _ = rs_keys[0]
        valu = sum(got[(kern, grid, "SQ_INSTS_VALU")])            # wavefront-level instructions, all launches
        salu = sum(got.get((kern, grid, "SQ_INSTS_SALU"), [0.0]))
        res["random_steps"] = {"valu_wave_insts": valu, "salu_wave_insts": salu,
                               "env_steps": child_note["random_env_steps"], "launches": child_note["random_launches"],
                               "valu_per_env_step": valu * 64 / child_note["random_env_steps"]}
    return res if len(res) > 1 else None


LINE_LIMIT = 4096                 # the final stdout line stays under this (a ~16 KiB line was dropped by the driver in round 5)
DETAIL_FILE = "bench_detail.json"


def _sig(x, digits=6):
    """Numbers of the compact line carry 6 significant digits (the full-precision record is bench_detail.json)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _pruned(d):
    return {k: v for k, v in d.items() if v is not None}


def compact_line(full, detail_path=DETAIL_FILE):
    """The ONE line the driver keeps: the contract's fields, the headline's roofline and cpu_baseline objects and, per
    secondary workload, {value, unit, bound / frac, what was parity-checked} — scalars only (plus per-rank arrays of
    n_gpus numbers at N > 1), no prose.  Everything else (notes, sources, per-position tables, quality curves, the
    roofline legs in full) is in `full`, which main() writes to bench_detail.json and prints as an EARLIER stdout line.
    tests/test_bench_line.py holds this under LINE_LIMIT characters on canned N = 1 and N = 8 records."""
    rf = full.get("roofline") or {}
    n = _get(full, "config", "states_per_gpu")
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = "env-steps/sec (batched LegalActions+ApplyAction+status, connect_four)"
    if (full.get("n_gpus") or 1) > 1:
        line["collective_backend"] = full.get("collective_backend")
        line["rccl_world"] = full.get("rccl_world")
    line["config"] = _pruned({
        "workload": f"connect_four fused LegalActions+ApplyAction+status, 2^{(n or 1).bit_length() - 1} states/GPU, seed 0x5EED",
        "states_per_gpu": n, "launches_per_step": _get(full, "config", "launches_per_step"),
        "env_steps_per_step": _get(full, "config", "env_steps_per_step"),
        "parallelism": f"{full.get('n_gpus')} independent shard(s), no collective"})
    line["roofline"] = _pruned({
        "bound": "hbm", "kernel": rf.get("kernel"), "achieved": rf.get("achieved"), "peak": rf.get("peak"),
        "unit": rf.get("unit"), "frac": rf.get("frac"), "traffic": rf.get("traffic"),
        "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"), "avg_launch_us": rf.get("avg_launch_us"),
        "launches_timed": rf.get("launches_timed"), "resident_in": rf.get("bound") if rf.get("bound") != "hbm" else None,
        "traffic_measured_in_this_run": rf.get("traffic_measured_in_this_run"),
        "copy_gbs": _get(rf, "copy_ceiling", "gbs"), "frac_of_copy": rf.get("frac_of_copy_ceiling"),
        "hbm_frac": rf.get("hbm_frac"), "hbm_achieved": rf.get("hbm_achieved"), "hbm_avg_launch_us": rf.get("hbm_avg_launch_us"),
        "hbm_states": rf.get("hbm_states"), "hbm_traffic": rf.get("hbm_traffic"),
        "hbm_frac_of_copy": rf.get("hbm_frac_of_copy_ceiling")})
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pruned({"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                        "kind": cb.get("kind"), "single_thread_value": cb.get("single_thread_value"),
                                        "sample": cb.get("sample_short") or (cb.get("sample") or "")[:90],
                                        "gpu_over_cpu": cb.get("gpu_over_cpu")})
    line["parity_checked_states"] = full.get("parity_checked_states", 0)
    line["parity_against"] = _get(full, "parity", "against")
    pr = full.get("per_rank")
    if pr:
        line["per_rank"] = {"env_steps_per_s": pr.get("env_steps_per_s"), "avg_launch_us": pr.get("avg_launch_us")}
    sec = full.get("secondary")
    if isinstance(sec, dict) and "error" in sec:
        line["secondary"] = {"error": str(sec["error"])[:200]}
    elif isinstance(sec, dict):
        out = {}
        m = sec.get("mcts") or {}
        if m:
            out["mcts"] = _pruned({
                "value": m.get("value"), "unit": m.get("unit"), "workload": "hex(9) 2^16 roots x 1024 sims",
                "bound": "scalar_issue" if _get(m, "roofline", "frac_of_scalar_issue_bound") is not None else None,
                "frac": _get(m, "roofline", "frac_of_scalar_issue_bound"),
                "mix_profile_current": _get(m, "roofline", "source_current"),
                "parity_checked_roots": m.get("parity_checked_roots"), "cpu_value": _get(m, "cpu_baseline", "value"),
                "cpu_cores": _get(m, "cpu_baseline", "cores"),
                "per_rank_sims_per_s": m.get("per_rank_sims_per_s"),
                "single_rank_all_roots": _get(m, "single_rank_all_roots", "value"),
                "strong_scaling_efficiency": m.get("strong_scaling_efficiency")})
        c = sec.get("cfr") or {}
        if c:
            out["cfr"] = _pruned({
                "value": c.get("value"), "unit": c.get("unit"), "workload": "kuhn_poker CFRSolver",
                "bound": "wave_issue" if _get(c, "roofline", "frac") is not None else None, "frac": _get(c, "roofline", "frac"),
                "mix_profile_current": _get(c, "roofline", "source_current"),
                "parity_checked_iterations": c.get("parity_checked_iterations"),
                "parity_max_rel_error": _get(c, "parity", "max_table_rel_error"),
                "cpu_value": _get(c, "cpu_baseline", "value"), "replicas_value": _get(c, "replicas", "value")})
            l2 = c.get("leduc") or {}
            if "value" in l2:
                out["cfr_leduc"] = _pruned({"value": l2.get("value"), "unit": l2.get("unit"),
                                            "parity_checked_iterations": l2.get("parity_checked_iterations"),
                                            "cpu_value": _get(l2, "cpu_baseline", "value")})
            elif "error" in l2:
                out["cfr_leduc"] = {"error": str(l2["error"])[:120]}
            l3 = c.get("leduc_3_players") or {}
            if l3:
                out["cfr_leduc_3p"] = _pruned({"value": l3.get("value"), "unit": l3.get("unit"), "bound": "hbm",
                                               "frac": l3.get("frac_of_hbm_peak"), "kernel": l3.get("kernel"),
                                               "parity_checked_iterations": l3.get("parity_checked_iterations"),
                                               "parity_max_rel_error": _get(l3, "parity", "max_table_rel_error"),
                                               "cpu_value": _get(l3, "cpu_baseline", "value")})
        x = sec.get("mccfr") or {}
        if x:
            ar = None
            if x.get("allreduce_us") is not None:
                key = "rccl" if "nccl" in str(x.get("allreduce_backend")) else str(x.get("allreduce_backend", "host")).split()[-1]
                ar = _pruned({key: x.get("allreduce_us"), "oneshot": _get(x, "oneshot", "allreduce_us")})
            out["mccfr"] = _pruned({
                "value": x.get("value"), "unit": x.get("unit"), "workload": "leduc_poker ES-MCCFR 16 x 2^20",
                "bound": "latency" if x.get("roofline") else None,
                "frac_of_scalar_issue": _get(x, "roofline", "frac_of_scalar_issue_bound"),
                "waves_per_simd": _get(x, "roofline", "waves_per_simd"),
                "mix_profile_current": _get(x, "roofline", "source_current"),
                "parity_checked_trajectories": x.get("parity_checked_trajectories"),
                "parity_max_error_over_tolerance": _get(x, "parity", "max_error_over_tolerance"),
                "nash_conv_after": x.get("nash_conv_after"), "cpu_value": _get(x, "cpu_baseline", "value"),
                "allreduce_us": ar, "allreduce_bytes": x.get("allreduce_bytes"),
                "oneshot_trajectories_per_s": _get(x, "oneshot", "trajectories_per_s"),
                "oneshot_error": str(_get(x, "oneshot", "error"))[:120] if _get(x, "oneshot", "error") else None,
                "tables_finite": x.get("tables_finite"),
                "seconds_to_nash_conv_0.1": _get(x, "quality", "seconds_to_nash_conv", "0.1", "seconds")})
        e = sec.get("env_step") or {}
        if "value" in e:
            out["env_step"] = _pruned({"value": e.get("value"), "unit": e.get("unit"), "bound": "infinity_cache",
                                       "frac": _get(e, "roofline", "frac_back_to_back"),
                                       "frac_per_launch_events": _get(e, "roofline", "frac"),
                                       "kernel_us": _get(e, "roofline", "kernel_us_back_to_back"),
                                       "compact_kernel_us": _get(e, "roofline", "compact", "kernel_us_back_to_back"),
                                       "compact_frac": _get(e, "roofline", "compact", "frac_back_to_back")})
        elif "error" in e:
            out["env_step"] = {"error": str(e["error"])[:120]}
        h = sec.get("hex_step") or {}
        if "value" in h:
            out["hex_step"] = _pruned({"value": h.get("value"), "unit": h.get("unit"), "states": h.get("states"),
                                       "bound": _get(h, "roofline", "bound"),
                                       "frac": _get(h, "roofline", "frac_on_bytes_moved"),
                                       "frac_on_survey_bytes": _get(h, "roofline", "frac"),
                                       "kernel_us": h.get("kernel_us_per_launch"),
                                       "parity_checked_states": h.get("parity_checked_states")})
        elif "error" in h:
            out["hex_step"] = {"error": str(h["error"])[:120]}
        pe = _get(sec, "policy_evaluation", "per_game") or {}
        if pe:
            out["nash_conv_us"] = _pruned({"kuhn": _get(pe, "kuhn_poker", "us_per_nash_conv"),
                                           "leduc": _get(pe, "leduc_poker", "us_per_nash_conv"),
                                           "leduc_3p": (_get(pe, "leduc_poker(players=3)", "ms_per_nash_conv") or 0) * 1e3 or None})
        t = sec.get("ttt_mcts") or {}
        if t:
            out["ttt_mcts"] = _pruned({"value": _get(t, "device_single_root", "value"), "unit": "sims/s",
                                       "cpu_value": _get(t, "cpu_baseline", "value"), "cpu_cores": _get(t, "cpu_baseline", "cores"),
                                       "error": (str(_get(t, "device_single_root", "error"))[:120]
                                                 if _get(t, "device_single_root", "error") else None)})
        line["secondary"] = out
    p = full.get("persistent")
    if p:
        line["persistent"] = _pruned({"value": p.get("value"), "unit": p.get("unit"), "k": p.get("k"),
                                      "bound": "vector_issue" if p.get("roofline") else None,
                                      "frac": _get(p, "roofline", "frac_of_vector_issue_bound")})
    line["detail"] = detail_path

    def rounded(o):
        if isinstance(o, dict):
            return {k: rounded(v) for k, v in o.items()}
        return _sig(o)
    line = rounded(line)
    # last resort, so that the driver's record never loses the headline: shed the secondary objects, widest first
    while len(json.dumps(line)) >= LINE_LIMIT and isinstance(line.get("secondary"), dict) and line["secondary"]:
        widest = max(line["secondary"], key=lambda k: len(json.dumps(line["secondary"][k])))
        del line["secondary"][widest]
        line["secondary_truncated"] = True
    return line


def read_lines(stdout):
    """(compact line, full record) out of a bench.py run's stdout: the compact line is the LAST line, the full record
    the earlier {"bench_detail": ...} line."""
    rows = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert rows and len(rows[-1]) < LINE_LIMIT, (len(rows), len(rows[-1]) if rows else 0)
    full = next((json.loads(ln)["bench_detail"] for ln in reversed(rows[:-1]) if ln.startswith('{"bench_detail"')), None)
    return json.loads(rows[-1]), full


def emit(full):
    """bench_detail.json (+ the same record as an earlier stdout line, wrapped so that it never looks like the bench
    line), then the compact line LAST."""
    detail_path = DETAIL_FILE
    for d in (os.environ.get("OSG_BENCH_DETAIL_DIR"), ROOT, os.getcwd()):
        if not d:
            continue
        try:
            with open(os.path.join(d, DETAIL_FILE), "w") as f:
                json.dump(full, f, indent=1)
            detail_path = os.path.relpath(os.path.join(d, DETAIL_FILE), ROOT) if d.startswith(ROOT) else os.path.join(d, DETAIL_FILE)
            break
        except OSError:
            continue
    print(json.dumps({"bench_detail": full}), flush=True)
    line = compact_line(full, detail_path)
    text = json.dumps(line, allow_nan=False)
    assert len(text) < LINE_LIMIT, len(text)
    print(text, flush=True)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20,
                    help=f"timed steps; one step = {LAUNCHES_PER_STEP} launches of the fused kernel over the batch")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--states", type=int, default=STATES_PER_GPU, help="states per GPU (default 2^20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the MCTS / CFR / MCCFR workloads")
    ap.add_argument("--no-legs", action="store_true", help="skip the DRAM-true and persistent legs")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the in-run rocprofv3 counter passes")
    ap.add_argument("--parity-states", type=int, default=-1,
                    help="states of rank 0's timed batch checked against the CPU reference after the timed region "
                         "(-1: all of them, 0: none)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and not args.pmc_child and (world_env is None or (world_env == "1" and "OSG_BENCH_SPAWNED" not in os.environ)):
        raise SystemExit(spawn_ranks(args.gpus))

    import datetime
    import torch
    import torch.distributed as dist
    import open_spiel_amd as osa

    if args.pmc_child:
        torch.cuda.set_device(0)
        pmc_child(torch, osa)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # OSG_DIST_BACKEND=gloo lets the N>1 code path run on a box with fewer GPUs than ranks (ranks
    # share devices, collectives go through the host): a test hook, never the measured configuration.
    backend = os.environ.get("OSG_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} needs {world} devices for RCCL (one rank per GPU) but this box has "
                         f"{torch.cuda.device_count()}; OSG_DIST_BACKEND=gloo exercises the code path on shared devices")
    device_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device_index)
    host_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        long_wait = datetime.timedelta(minutes=30)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=long_wait)
            # long waits (rank 0 timing the CPU reference) block on a socket here instead of spinning on the GPU
            host_group = dist.new_group(backend="gloo", timeout=long_wait)
        else:
            dist.init_process_group(backend, timeout=long_wait)

    def host_barrier():
        if world > 1:
            dist.barrier(group=host_group) if host_group is not None else dist.barrier()

    def gather_floats(x):
        """[x of rank 0, x of rank 1, ...] on every rank."""
        if world == 1:
            return [float(x)]
        t = torch.zeros(world, dtype=torch.float64, device="cuda")
        t[rank] = x
        dist.all_reduce(t)
        return [float(v) for v in t.tolist()]

    ctx = osa.Context(device_index)
    n = args.states
    src, actions = synth_batch(osa, torch, ctx, n, SEED, rank * n)
    dst = osa.StateBatch(ctx, "connect_four", n)
    mask, status = src.step_buffers()

    def one_launch():
        src.step(actions, dst=dst, mask=mask, status=status)

    def one_step():
        for _ in range(LAUNCHES_PER_STEP):
            one_launch()

    for _ in range(args.warmup):
        one_step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region: exactly K steps (K x LAUNCHES_PER_STEP launches), barrier + synchronize on both
    # sides.  Two HIP events on the launching stream (the context is bound to torch's current stream)
    # bracket the same launches: with the stream saturated, event time / launches is the kernel's
    # average launch duration, the figure rocprofv3 --stats reports as well.
    launches = args.steps * LAUNCHES_PER_STEP
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        one_step()
    ev1.record()
    fence()
    elapsed_local = time.perf_counter() - t0
    avg_kernel_s = ev0.elapsed_time(ev1) / 1e3 / launches
    per_rank_elapsed = gather_floats(elapsed_local)
    per_rank_kernel_us = gather_floats(avg_kernel_s * 1e6)
    elapsed = max(per_rank_elapsed)
    assert int((status & 0x40).sum().item()) == 0, "synthetic actions must all be legal"
    # parity of the TIMED launches: every state of rank 0's shard (or --parity-states of them) regenerated by the CPU
    # reference and compared with what the last timed launch wrote (the checker never runs inside the timed region)
    parity = None
    if rank == 0 and args.parity_states != 0:
        want_states = n if args.parity_states < 0 else min(n, args.parity_states)
        parity = parity_check(torch, src, dst, actions, mask, status, SEED, rank * n, want_states)

    legs = None
    if not args.no_legs and rank == 0:
        try:
            csecs, cbytes = copy_ceiling(osa, torch, ctx, n, 1000, 100)
            legs = {"copy": (csecs, cbytes), "persistent": persistent_leg(osa, torch, ctx, src, rank)}
            free_b, _total = torch.cuda.mem_get_info()
            if free_b > 4 * ALGO_BYTES_PER_STEP * DRAM_LEG_STATES:
                legs["dram"] = dram_leg(osa, torch, ctx, src, actions)
        except Exception as e:  # noqa: BLE001 - the headline is measured already; say what failed and go on
            print(f"[bench] roofline legs failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            legs = None
    host_barrier()

    secondary = None
    if not args.no_secondary:
        del src, dst
        # The headline above is measured; whatever happens in the workloads beside it must not cost the line.
        try:
            secondary = secondary_workloads(osa, torch, dist, ctx, rank, world, with_cpu=not args.no_cpu_baseline,
                                            host_barrier=host_barrier, gather_floats=gather_floats)
        except Exception as e:  # noqa: BLE001 - reported in the line, never swallowed silently
            import traceback
            secondary = {"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-2000:]}
            print(f"[bench] secondary workloads failed on rank {rank}: {secondary['error']}", file=sys.stderr, flush=True)

    if rank == 0:
        traffic, traffic_source, dram_traffic = pmc_traffic()
        live = None if (args.no_pmc or n != STATES_PER_GPU) else measure_pmc()
        if live and "step" in live:
            traffic, traffic_source = live["step"]["bytes_per_launch"], live["source"]
            dram_traffic = live.get("dram", {}).get("bytes_per_launch", dram_traffic)
        total_env_steps = n * world * launches
        value = total_env_steps / elapsed
        achieved = ALGO_BYTES_PER_STEP * n / avg_kernel_s / 1e9
        roofline = {"bound": "infinity_cache" if ALGO_BYTES_PER_STEP * n < (200 << 20) else "hbm",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                    "traffic_measured_in_this_run": bool(live and "step" in live),
                    "kernel": "k_step_c4std2", "algorithmic_bytes_per_launch": ALGO_BYTES_PER_STEP * n,
                    "avg_launch_us": avg_kernel_s * 1e6, "launches_timed": launches,
                    "peak_note": "peak = HBM3E spec 8 TB/s (the denominator BASELINE.json names); one launch moves "
                                 f"{ALGO_BYTES_PER_STEP * n / 1e6:.1f} MB, which stays resident in the 256 MiB Infinity "
                                 "Cache between launches, hence bound = infinity_cache: `frac` at the config's 2^20 states is Infinity-Cache "
                                 "bandwidth over the HBM denominator; hbm_frac (= dram_leg.frac, 2^24 states) is the HBM-true figure"}
        if legs is not None:
            csecs, cbytes = legs["copy"]
            roofline["copy_ceiling"] = {"gbs": cbytes / csecs / 1e9, "us": csecs * 1e6, "bytes": cbytes,
                                        "what": "osg_copy_bytes: uint4 copy (non-temporal stores) of the same number of bytes, same stream"}
            roofline["frac_of_copy_ceiling"] = achieved / (cbytes / csecs / 1e9)
            if "dram" in legs:
                roofline["dram_leg"] = legs["dram"]
                roofline["dram_leg"]["traffic"] = dram_traffic
                # the HBM-true figures as scalars beside `frac` (a record that keeps only this object's top level must
                # still show them): the same kernel over 2^24 states, every byte from / to HBM
                roofline["hbm_frac"] = legs["dram"]["frac"]
                roofline["hbm_achieved"] = legs["dram"]["achieved"]
                roofline["hbm_avg_launch_us"] = legs["dram"]["avg_launch_us"]
                roofline["hbm_states"] = legs["dram"]["states"]
                roofline["hbm_traffic"] = dram_traffic
                roofline["hbm_frac_of_copy_ceiling"] = legs["dram"].get("frac_of_copy_ceiling")
        line = {
            "metric": "env-steps/sec (batched LegalActions+ApplyAction+status, connect_four)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic", "collective_backend": backend if world > 1 else None,
            # what the process group itself reports (the first real N > 1 run must show RCCL carried it)
            "rccl_world": (dist.get_world_size() if (world > 1 and dist.get_backend() == "nccl") else None),
            "config": {"workload": f"connect_four fused step, {n} states/GPU (2^{n.bit_length() - 1}), "
                                   f"out-of-place SoA bitboards, seed 0x5EED; one step = {LAUNCHES_PER_STEP} "
                                   f"back-to-back launches over the batch ({LAUNCHES_PER_STEP} x {n} env-steps per GPU)",
                       "states_per_gpu": n, "launches_per_step": LAUNCHES_PER_STEP,
                       "env_steps_per_step": n * world * LAUNCHES_PER_STEP,
                       "parallelism": f"{world} independent shard(s), no collective"},
            "roofline": roofline,
            "parity_checked_states": parity["states"] if parity else 0,
            "parity": parity,
        }
        if world > 1:
            line["per_rank"] = {"env_steps_per_s": [n * launches / t for t in per_rank_elapsed],
                                "avg_launch_us": per_rank_kernel_us,
                                "note": "value = all ranks' env-steps / the slowest rank's time (barrier + synchronize on both sides)"}
        if legs is not None:
            line["persistent"] = legs["persistent"]
            rs = (live or {}).get("random_steps") or committed_random_steps_mix()
            if rs:
                # issue-rate view, like the search kernel's: wave-level vector instructions per SIMD x the
                # measured issue interval of the cheapest vector instruction (tools/clock_probe.hip: 1.03 ns)
                p = line["persistent"]
                simds = torch.cuda.get_device_properties(0).multi_processor_count * 4
                valu_per_step = rs["valu_per_env_step"]
                issue_s = p["env_steps"] * valu_per_step / 64 / simds * 1.03e-9
                p["roofline"] = {"bound": "vector-unit instruction issue", "valu_per_env_step": valu_per_step,
                                 "vector_issue_seconds_at_least": issue_s, "seconds": p["seconds"],
                                 "frac_of_vector_issue_bound": issue_s / p["seconds"],
                                 "source": rs.get("source", "in-run rocprofv3 --pmc SQ_INSTS_VALU pass"),
                                 "note": "a wave64 vector instruction issues every 1.03 ns per SIMD at best (64-bit "
                                         "shifts, multiplies: 1.8-1.9 ns), so the fraction is a lower bound of the "
                                         "vector unit's busy share",
                                 }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
            if world > 1:
                line["cpu_baseline"]["note"] = "timed on rank 0's host while the other ranks wait on a socket barrier"
        if secondary is not None:
            line["secondary"] = secondary
        emit(line)
    host_barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
