"""Config 5 as bench.py times it: leduc ES-MCCFR, 16 mini-batches of 2^20 trajectories with the fold after each (the
policy moves: a launch on a trained table is longer than on the uniform one), total and per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
from open_spiel_amd import _abi
if os.environ.get("OSG_VARIANT_LIB"): _abi.LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"])
import open_spiel_amd.distributed as osd
ctx = osa.Context(0)
solver = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
sharded = osd.ShardedMccfr(solver)
sharded.run_minibatch(7, 1 << 12); ctx.synchronize()
for rep in range(2):
    per = []
    t0 = time.perf_counter()
    for k in range(16):
        t1 = time.perf_counter()
        sharded.run_minibatch(7 + rep, 1 << 20)
        ctx.synchronize()
        per.append((time.perf_counter() - t1) * 1e6)
    dt = time.perf_counter() - t0
    print(f"16 mini-batches of 2^20: {16 * (1 << 20) / dt:.3e} trajectories/s; per launch (us): first {per[0]:.0f}, last {per[-1]:.0f}, mean {sum(per) / 16:.0f}; nash_conv {solver.nash_conv():.4f}", flush=True)
