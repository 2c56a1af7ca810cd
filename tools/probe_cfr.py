import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_spiel_amd import _abi
if os.environ.get("OSG_VARIANT_LIB"): _abi.LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"])
import time, torch, open_spiel_amd as osa
ctx = osa.Context(0)
for game, iters in [("kuhn_poker", 20000), ("kuhn_poker(players=3)", 2000), ("leduc_poker", 500)]:
    s = osa.TabularSolver(ctx, game)
    s.evaluate_and_update_policy(10); torch.cuda.synchronize()
    t = time.time(); s.evaluate_and_update_policy(iters); torch.cuda.synchronize(); dt = time.time() - t
    print(game, "CFR iters/s", iters / dt, "us/iter", dt / iters * 1e6, flush=True)
# leduc CFR: one workgroup per deal subtree (auto) beside the single-workgroup path kernel and the full-grid phases
for label, kw, iters in [("split (auto)", {}, 20000), ("split, 1 iteration per launch", {}, 0), ("path (one workgroup)", dict(general_kernel="path"), 500),
                         ("grid phases", dict(general_kernel="grid"), 200)]:
    s = osa.TabularSolver(ctx, "leduc_poker", **kw)
    s.evaluate_and_update_policy(10); torch.cuda.synchronize()
    if iters:
        t = time.time(); s.evaluate_and_update_policy(iters); torch.cuda.synchronize(); dt = time.time() - t
    else:
        iters = 2000
        t = time.time()
        for _ in range(iters): s.evaluate_and_update_policy(1)
        torch.cuda.synchronize(); dt = time.time() - t
    print(f"leduc_poker CFR [{label}] iters/s {iters / dt:.1f} us/iter {dt / iters * 1e6:.2f} nash_conv {s.nash_conv():.6f}", flush=True)
s = osa.TabularSolver(ctx, "leduc_poker(players=3)")
s.evaluate_and_update_policy(2); torch.cuda.synchronize()
t = time.time(); s.evaluate_and_update_policy(20); torch.cuda.synchronize(); dt = time.time() - t
print(f"leduc_poker(players=3) CFR [grid phases] iters/s {20 / dt:.1f} us/iter {dt / 20 * 1e6:.1f}", flush=True)
for game, n in [("kuhn_poker", 1 << 20), ("leduc_poker", 1 << 20), ("leduc_poker", 1 << 22)]:
    s = osa.TabularSolver(ctx, game, mccfr=True)
    s.run_mccfr(1, 4096); torch.cuda.synchronize()
    t = time.time(); s.run_mccfr(1, n, first_trajectory=4096); torch.cuda.synchronize(); dt = time.time() - t
    print(game, "MCCFR traj/s", n / dt, "ms", dt * 1e3, flush=True)
