"""The fused hex step (no mask row) on the boards other than hex(9), 2^22 states: microseconds per launch and the
fraction of 8 TB/s on the bytes moved (record in + record out + action + status).  OSG_HEX_FOLD=0 keeps the separate
meta word."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
n = 1 << 22
for game in ("hex(board_size=5)", "hex(board_size=7)", "hex(board_size=8)", "hex"):   # (11 x 11 is the default)
    b = osa.StateBatch(ctx, game, n); b.random_steps(3, 12)
    dst = osa.StateBatch(ctx, game, n)
    lm = b.legal_actions_mask()
    acts = torch.where(lm.bool().any(1), (lm.to(torch.float32) * torch.rand(lm.shape, device="cuda")).argmax(1),
                       torch.full((n,), 255, device="cuda")).to(torch.uint8)
    del lm
    small = b.desc.num_distinct_actions <= 128
    mask, status = b.step_buffers()
    def go():
        if small: b.step(acts, dst=dst, status=status, want_mask=False)
        else: b.step(acts, dst=dst, mask=mask, status=status)
    for _ in range(5): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    rec = b.desc.state_words * 4
    moved = 2 * rec + 2 + (0 if small else b.desc.compact_mask_bytes)
    print(f"{game:22s} {b.desc.state_words:3d} words  {us:8.1f} us per 2^22 states  {moved * n / us / 1e6 / 8:.3f} of 8 TB/s on {moved} B moved", flush=True)
    del b, dst, acts, mask, status
