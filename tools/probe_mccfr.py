"""ES-MCCFR kernel timings: the LDS-resident traversal vs the general kernel (osg_cfr_cfg.kernel = 1),
on a fresh table and on a table that has already seen 16 mini-batches (non-uniform policies)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, open_spiel_amd as osa
ctx = osa.Context(0)
for kind, game, n in [("external", "kuhn_poker", 1 << 20), ("external", "leduc_poker", 1 << 20),
                      ("external", "leduc_poker", 1 << 22), ("external", "leduc_poker", 4096), ("external", "leduc_poker", 1 << 17),
                      ("external", "kuhn_poker(players=3)", 1 << 20), ("outcome", "kuhn_poker", 1 << 22),
                      ("outcome", "leduc_poker", 1 << 22)]:
    for general in (True, False):
        s = osa.TabularSolver(ctx, game, mccfr=kind, general_kernel=general)
        s.run_mccfr(1, 4096); torch.cuda.synchronize()
        t = time.time(); s.run_mccfr(1, n, first_trajectory=4096); torch.cuda.synchronize(); dt = time.time() - t
        for b in range(16):
            s.run_mccfr(2, 1 << 16, first_trajectory=b << 16)
        torch.cuda.synchronize()
        t = time.time(); s.run_mccfr(3, n); torch.cuda.synchronize(); dt2 = time.time() - t
        print(f"{kind} {game} n={n} {'general' if general else 'resident'}: fresh {n / dt:.3e} traj/s ({dt * 1e3:.3f} ms); "
              f"trained {n / dt2:.3e} traj/s ({dt2 * 1e3:.3f} ms)", flush=True)
