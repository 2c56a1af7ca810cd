"""leduc_poker ES-MCCFR: what a rank's mini-batch step costs when the 2^14-trajectory mini-batch (the size that
converges best, profiles/r02_mccfr_quality.log) is sharded over N ranks — sample 2^14 / N trajectories, fold — on ONE
GPU (the exchange is not in it: the one-shot all-reduce of the 44 928-byte deltas is 5.5 us on one device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
s = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
for _ in range(50): s.mccfr_sample(7, 1 << 14); s.mccfr_apply_deltas()
for ranks in (1, 2, 4, 8, 16):
    per = (1 << 14) // ranks
    ctx.synchronize(); t0 = time.perf_counter()
    for k in range(300):
        s.mccfr_sample(9, per, first_trajectory=k * (1 << 14))
        s.mccfr_apply_deltas()
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 300
    print(f"ranks {ranks:2d}: {per:6d} trajectories per rank and mini-batch: {dt * 1e6:7.1f} us per step (sample + fold)", flush=True)
if os.environ.get("OSG_MCCFR_STAMPS"):
    for per in (1 << 14, 1 << 11, 64):
        s.mccfr_sample(11, per); s.mccfr_apply_deltas(); ctx.synchronize()
