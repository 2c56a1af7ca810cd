"""Fused step of the small games with n even (the vectorised kernel: V states per thread) and n odd (one state
per thread): which layout of the work is faster at which size."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for game, depth, B in [("leduc_poker", 4, 35), ("leduc_poker(players=3)", 5, 35)]:
    for n in ((1 << 24), (1 << 24) + 1, (1 << 20), (1 << 20) + 1, (1 << 24), (1 << 24) + 1):
        b = osa.StateBatch(ctx, game, n); b.random_steps(3, depth)
        dst = osa.StateBatch(ctx, game, n)
        mask, status = b.step_buffers()
        lm = b.legal_actions_mask()
        acts = torch.where(lm.any(1), lm.to(torch.float32).argmax(1), torch.full((n,), 255, device="cuda")).to(torch.uint8)
        del lm
        t = timeit(lambda: b.step(acts, dst=dst, mask=mask, status=status))
        print(f"{game} n={n}: {t*1e6:.1f} us  {B*n/t/8e12:.3f} of 8 TB/s", flush=True)
        del b, dst, mask, status, acts
