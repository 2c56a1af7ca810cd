"""hex(9) fused step at batch sizes around 2^24: does the power-of-two plane stride (13 + 13 plane streams 64 MiB apart:
every stream of a wavefront in the same memory channel) cost bandwidth?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
def timeit(fn, iters=20, warm=4):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
game = os.environ.get("PROBE_GAME", "hex(board_size=9)")
for n in [1 << 24, (1 << 24) + (1 << 14), (1 << 24) + 512 * 37, (1 << 24) - (1 << 20) + 2 * 1031, 3 * (1 << 22) + 64 * 11, 1 << 23, (1 << 23) + 2 * 4099]:
    b = osa.StateBatch(ctx, game, n); b.random_steps(3, 30)
    dst = osa.StateBatch(ctx, game, n)
    mask, status = b.step_buffers()
    lm = b.legal_actions_mask()
    acts = torch.where(lm.any(1), (lm.to(torch.float32) * torch.rand(lm.shape, device="cuda")).argmax(1),
                       torch.full((n,), 255, device="cuda")).to(torch.uint8)
    del lm
    s = timeit(lambda: b.step(acts, dst=dst, mask=mask, status=status))
    d = b.desc
    bytes_per = 2 * d.state_words * d.state_word_bytes + 1 + d.compact_mask_bytes + 1
    print(f"{game} n = {n:9d} (plane stride {n * d.state_word_bytes / 2**20:8.3f} MiB): {s * 1e6:8.1f} us  {bytes_per * n / s / 8e12:.3f} of 8 TB/s", flush=True)
    del b, dst, mask, status, acts
