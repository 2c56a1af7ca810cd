"""hex(9) MCTS, BASELINE config 4 (2^16 roots x 1024 simulations) and the 2^13-root shard an 8-GPU run gives each rank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_spiel_amd import _abi
if os.environ.get("OSG_VARIANT_LIB"): _abi.LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"])
import torch, open_spiel_amd as osa, bench
ctx = osa.Context(0)
for n in (1 << 13, 1 << 16):
    roots = bench.hex_roots(osa, torch, ctx, n, 0)
    best = None
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = roots.mcts_search(uct_c=2.0, max_simulations=1024, n_rollouts=1, seed=bench.SEED + rep)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    sims = float(res["root_stats"][:, 3].sum())
    print(f"hex(9) {n} roots x 1024 sims: {best:.4f} s  {sims / best:.4g} sims/s", flush=True)
    del roots
