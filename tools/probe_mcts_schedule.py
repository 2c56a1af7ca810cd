"""hex(9) MCTS (config 4): schedules of the wave-per-root search (OSG_MCTS_SCHEDULE, read at every launch) on the
2^16-root batch and on the 2^13-root shard one of 8 ranks gets.  Every schedule must give the same statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa, bench
ctx = osa.Context(0)
SCHEDULES = sys.argv[1:] or ["static", "queue:7", "lpt:7", "lpt:6", "lpt:5", "lpt:4", "queue:4", "lpt:8", "static"]
rates = {}
for n in (1 << 13, 1 << 16):
    roots = bench.hex_roots(osa, torch, ctx, n, 0)
    want = None
    for sched in SCHEDULES:
        os.environ["OSG_MCTS_SCHEDULE"] = sched
        best = None
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = roots.mcts_search(uct_c=2.0, max_simulations=1024, n_rollouts=1, seed=bench.SEED)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if rep:  # the first call sizes the pool / compiles nothing but warms the caches
                best = dt if best is None or dt < best else best
        sims = float(res["root_stats"][:, 3].sum())
        key = (int(res["child_visits"].to(torch.int64).sum()), float(res["child_reward"].sum()), int(res["best_action"].to(torch.int64).sum()))
        want = want or key
        rates[(n, sched)] = sims / best
        print(f"hex(9) {n:6d} roots x 1024 sims  {sched:9s} {best * 1e3:8.3f} ms  {sims / best:.4g} sims/s  "
              f"{'same statistics' if key == want else 'DIFFERENT STATISTICS ' + str(key) + ' vs ' + str(want)}", flush=True)
    del roots
for sched in SCHEDULES:
    a, b = rates.get((1 << 13, sched)), rates.get((1 << 16, sched))
    if a and b:
        print(f"{sched:9s} 2^13-root shard at {a / b:.3f} of its own 2^16 rate, {a / max(rates[(1 << 16, s)] for s in SCHEDULES):.3f} of the best 2^16 rate")
