"""A/B of the connect_four fused step over library variants (tools/build_variant.sh with SRC=osg_kernels):
HIP-event time per launch at 2^20 and 2^24 states, each library in its own process."""
import glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.dirname(HERE))
from open_spiel_amd import _abi
_abi.LIB_PATH = LIB
import torch, open_spiel_amd as osa, bench
ctx = osa.Context(0)
src, actions = bench.synth_batch(osa, torch, ctx, 1 << 20, bench.SEED, 0)
dst = osa.StateBatch(ctx, "connect_four", src.n)
mask, status = src.step_buffers()
best = min(bench.timed_launches(torch, lambda: src.step(actions, dst=dst, mask=mask, status=status), 2000, 200) for _ in range(3))
chk = int(mask.to(torch.int64).sum().item()) * 1000003 + int(status.to(torch.int64).sum().item())
leg = bench.dram_leg(osa, torch, ctx, src, actions)
print(f"{os.path.basename(LIB):28s} 2^20: {best * 1e6:.3f} us/launch ({35 * src.n / best / 8e12:.3f} of 8 TB/s)   2^24: {leg['avg_launch_us']:.2f} us ({leg['frac']:.3f})   checksum {chk}", flush=True)
'''
libs = sys.argv[1:] or sorted(glob.glob(os.path.join(HERE, "variants", "libosg_*.so")))
for lib in libs:
    code = f"HERE={HERE!r}\nLIB={os.path.abspath(lib)!r}\n" + CHILD
    subprocess.run([sys.executable, "-c", code], check=False)
