"""hex(9) fused step: states per thread x non-temporal stores (OSG_HEX_STEP, read at every launch) at 2^20, 2^22 and
2^24 states.  Bytes per step as tools/probe_kernels.py counts them: 2 x 52 state + 1 action + 12 mask + 1 status = 118."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_spiel_amd import _abi
if os.environ.get("OSG_VARIANT_LIB"): _abi.LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"])
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
def timeit(fn, iters, warm):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
game = "hex(board_size=9)"
for logn in (20, 22, 24):
    n = 1 << logn
    b = osa.StateBatch(ctx, game, n); b.random_steps(3, 30)
    dst = osa.StateBatch(ctx, game, n)
    mask, status = b.step_buffers()
    lm = b.legal_actions_mask()
    acts = torch.where(lm.any(1), (lm.to(torch.float32) * torch.rand(lm.shape, device="cuda")).argmax(1),
                       torch.full((n,), 255, device="cuda")).to(torch.uint8)
    del lm
    want = None
    for knob in ("1:0", "2:0", "2:1"):
        os.environ["OSG_HEX_STEP"] = knob
        s = timeit(lambda: b.step(acts, dst=dst, mask=mask, status=status), 100 if logn < 24 else 30, 10)
        key = (int(mask.to(torch.int64).sum()), int(status.to(torch.int64).sum()), int(torch.from_numpy(dst.raw_words().astype("int64")).sum()) if logn == 20 else 0)
        want = want or key
        print(f"hex(9) step n=2^{logn} states/thread:nt={knob}  {s * 1e6:9.2f} us  {118 * n / s / 8e12:.3f} of 8 TB/s  "
              f"{'same' if key == want else 'DIFFERENT'}", flush=True)
    # round 5: the same step without the successor's mask row (d_mask == NULL: 106 B moved per state + 3 B of SURVEY's
    # padding = its 109 B; the fraction is quoted on the bytes actually moved)
    os.environ.pop("OSG_HEX_STEP", None)
    s = timeit(lambda: b.step(acts, dst=dst, mask=mask, status=status), 100 if logn < 24 else 30, 10)
    print(f"hex(9) step n=2^{logn} default, mask row written      {s * 1e6:9.2f} us  {118 * n / s / 8e12:.3f} of 8 TB/s (118 B)", flush=True)
    s = timeit(lambda: b.step(acts, dst=dst, status=status, want_mask=False), 100 if logn < 24 else 30, 10)
    print(f"hex(9) step n=2^{logn} default, NO mask row           {s * 1e6:9.2f} us  {106 * n / s / 8e12:.3f} of 8 TB/s (106 B)", flush=True)
    del b, dst, mask, status, acts
