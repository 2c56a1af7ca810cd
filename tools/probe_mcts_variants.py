"""A/B of kernel variants: runs tools/probe_mcts_bench.py's measurement once per library under tools/variants/
(each in its own process, the library path swapped in before the first ABI call)."""
import glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.dirname(HERE))
from open_spiel_amd import _abi
_abi.LIB_PATH = LIB
import torch, open_spiel_amd as osa, bench
ctx = osa.Context(0)
for n in NS:
    roots = bench.hex_roots(osa, torch, ctx, n, 0)
    best = None
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = roots.mcts_search(uct_c=2.0, max_simulations=1024, n_rollouts=1, seed=bench.SEED + rep)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    sims = float(res["root_stats"][:, 3].sum())
    chk = int(res["child_visits"].to(torch.int64).mul(torch.arange(res["child_visits"].shape[1], device=res["child_visits"].device) + 1).sum())
    print(f"{os.path.basename(LIB):28s} hex(9) {n:6d} roots x 1024 sims: {best:.4f} s  {sims / best:.4g} sims/s  checksum {chk}", flush=True)
    del roots
'''
libs = sys.argv[1:] or sorted(glob.glob(os.path.join(HERE, "variants", "libosg_*.so")))
for lib in libs:
    code = f"HERE={HERE!r}\nLIB={os.path.abspath(lib)!r}\nNS=(1 << 13, 1 << 16)\n" + CHILD
    subprocess.run([sys.executable, "-c", code], check=False)
