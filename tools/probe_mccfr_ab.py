"""ES-MCCFR throughput (leduc_poker, mini-batches of 2^20 and 2^16) and step time (2^14) over library variants, each in its
own process: the product library against tools/variants/libosg_*.so."""
import glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.dirname(HERE))
from open_spiel_amd import _abi
if LIB: _abi.LIB_PATH = LIB
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
s = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
for _ in range(30): s.run_mccfr(3, 1 << 14)
out = []
for n, reps in ((1 << 20, 16), (1 << 17, 32), (1 << 16, 64), (1 << 14, 200)):
    ctx.synchronize(); t0 = time.perf_counter()
    for k in range(reps): s.run_mccfr(5, n, first_trajectory=k * n)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / reps
    out.append(f"2^{n.bit_length() - 1}: {dt * 1e6:8.1f} us = {n / dt:.3e} traj/s")
print(f"{NAME:12s} " + "   ".join(out), flush=True)
'''
libs = [("product", "")] + [(os.path.basename(p)[7:-3], p) for p in sorted(glob.glob(os.path.join(HERE, "variants", "libosg_*mccfr*.so")))]
for rep in range(2):
    for name, lib in libs:
        subprocess.run([sys.executable, "-c", f"HERE={HERE!r}\nLIB={lib!r}\nNAME={name!r}\n" + CHILD], check=False)
