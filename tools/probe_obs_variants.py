"""Tensor-pack kernels per library variant (tools/build_variant.sh with SRC=osg_kernels): python tools/probe_obs_variants.py lib1.so lib2.so"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.dirname(HERE))
from open_spiel_amd import _abi
_abi.LIB_PATH = LIB
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for game, depth, n in [("connect_four", 12, 1 << 20), ("connect_four", 12, 1 << 24), ("tic_tac_toe", 3, 1 << 24), ("hex(board_size=9)", 30, 1 << 18)]:
    b = osa.StateBatch(ctx, game, n); b.random_steps(3, depth)
    out = torch.empty((n, b.desc.obs_size), dtype=torch.float32, device="cuda")
    t = min(timeit(lambda: b.observation_tensor(0, out=out)) for _ in range(2))
    nbytes = n * (b.desc.state_words * b.desc.state_word_bytes + 4 * b.desc.obs_size)
    print(f"{os.path.basename(LIB):24s} {game} [{n}, {b.desc.obs_size}]: {t * 1e6:.1f} us  {nbytes / t / 8e12:.3f} of 8 TB/s", flush=True)
    del b, out
'''
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", f"HERE={HERE!r}\nLIB={os.path.abspath(lib)!r}\n" + CHILD], check=False)
