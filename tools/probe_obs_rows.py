"""Short-row tensor kernels: the wavefront-span form (k_observation_rows) beside the granule form
(k_observation_granules, OSG_OBS_ROWS=gran[:rounds], read at every launch) at 2^20 and 2^24 states; outputs compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
def timeit(fn, iters, warm):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
forms = sys.argv[1:] or ["rows", "gran", "gran:2", "gran:4"]
for game, depth in [("tic_tac_toe", 3), ("kuhn_poker", 2), ("leduc_poker", 4)]:
    for logn in (20, 24):
        n = 1 << logn
        b = osa.StateBatch(ctx, game, n); b.random_steps(3, depth)
        d = b.desc
        sb = d.state_words * d.state_word_bytes
        for which, size in (("obs", d.obs_size), ("info", d.info_size)):
            if not size: continue
            out = torch.empty((n, size), dtype=torch.float32, device="cuda")
            fn = (lambda: b.observation_tensor(0, out=out)) if which == "obs" else (lambda: b.information_state_tensor(0, out=out))
            want = None
            for form in forms:
                os.environ["OSG_OBS_ROWS"] = form
                out.fill_(-7.0)
                s = timeit(fn, 50 if logn == 20 else 20, 5)
                same = "" if want is None else ("same" if torch.equal(out, want) else "DIFFERENT")
                if want is None: want = out.clone()
                print(f"{game} {which} [2^{logn},{size}] {form:7s} {s * 1e6:9.2f} us  {n * (sb + 4 * size) / s / 8e12:.3f} of 8 TB/s  {same}", flush=True)
            del out, want
        del b
