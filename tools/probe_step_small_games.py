"""Fused step of the other games at two sizes for a library variant: python tools/probe_step_small_games.py <lib.so>"""
import os, sys
sys.path.insert(0, "/root/repo")
from open_spiel_amd import _abi
_abi.LIB_PATH = sys.argv[1]
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for game, depth, B in [("hex(board_size=9)", 30, 109), ("leduc_poker", 4, 35), ("kuhn_poker", 2, 19)]:
    for n in ((1 << 20), (1 << 22) + 1):
        b = osa.StateBatch(ctx, game, n); b.random_steps(3, depth)
        dst = osa.StateBatch(ctx, game, n)
        mask, status = b.step_buffers()
        lm = b.legal_actions_mask()
        acts = torch.where(lm.any(1), lm.to(torch.float32).argmax(1), torch.full((n,), 255, device="cuda")).to(torch.uint8)
        del lm
        t = min(timeit(lambda: b.step(acts, dst=dst, mask=mask, status=status)) for _ in range(2))
        print(f"{os.path.basename(sys.argv[1]):22s} {game} n=2^{n.bit_length()-1}: {t*1e6:.1f} us  {B*n/t/8e12:.3f} of 8 TB/s", flush=True)
        del b, dst, mask, status, acts
