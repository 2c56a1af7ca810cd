import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_spiel_amd import _abi
if os.environ.get("OSG_VARIANT_LIB"): _abi.LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"])
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
for game, n in [("kuhn_poker", 1 << 20), ("kuhn_poker(players=3)", 1 << 20), ("kuhn_poker(players=5)", 1 << 20), ("leduc_poker", 1 << 20),
                ("leduc_poker(suit_isomorphism=True)", 1 << 20), ("leduc_poker(players=3)", 1 << 20)]:
    try:
        s = osa.TabularSolver(ctx, game, mccfr=True)
        s.run_mccfr(1, 4096); torch.cuda.synchronize()
        for rep in range(2):
            t = time.time(); s.run_mccfr(1 + rep, n, first_trajectory=4096 + rep * n); torch.cuda.synchronize(); dt = time.time() - t
        print(f"{game:40s} ES-MCCFR {n / dt:.3e} traj/s  ({dt * 1e3:.2f} ms per 2^20)  [{s.last_kernel()}]", flush=True)
        del s
    except Exception as e:
        print(game, "ERROR", e, flush=True)
