"""A/B of the tensor-pack kernel forms: OSG_OBS_FORM=0 (span per wavefront, the round-3 kernels), 1 (one aligned
16-byte piece per thread, plain stores), 2 (same, non-temporal stores: the default), each in its own process (the
switch is read once), at DRAM-true sizes (2^24 states; hex(9): 2^22) or PROBE_SHIFT=k smaller by 2^k.  Every form's
output is compared with form 0's (first 2^18 states and the last 4096 rows, bit for bit) before it is timed."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, hashlib
sys.path.insert(0, os.environ["OSG_ROOT"])
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
def timeit(fn, iters=int(os.environ.get('PROBE_ITERS', '20')), warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for game, depth, n in [("connect_four", 12, 1 << 24), ("tic_tac_toe", 3, 1 << 24), ("hex(board_size=9)", 30, 1 << 22),
                       ("kuhn_poker", 2, 1 << 24), ("leduc_poker", 4, 1 << 24)]:
    if os.environ.get("PROBE_GAMES") and game.split("(")[0] not in os.environ["PROBE_GAMES"].split(","): continue
    n >>= int(os.environ.get("PROBE_SHIFT", "0"))
    b = osa.StateBatch(ctx, game, n); b.random_steps(3, depth)
    d = b.desc
    sb = d.state_words * d.state_word_bytes
    for which, size in ((0, d.obs_size), (1, d.info_size)):
        if not size: continue
        out = torch.empty((n, size), dtype=torch.float32, device="cuda")
        fn = (lambda: b.observation_tensor(0, out=out)) if which == 0 else (lambda: b.information_state_tensor(0, out=out))
        fn(); torch.cuda.synchronize()
        digest = hashlib.sha256(out[: 1 << 18].cpu().numpy().tobytes()).hexdigest()[:16]
        tail = hashlib.sha256(out[-4096:].cpu().numpy().tobytes()).hexdigest()[:16]
        s = timeit(fn)
        print(json.dumps({"game": game, "which": which, "shape": [n, size], "us": s * 1e6,
                          "frac_of_8TBs": n * (sb + 4 * size) / s / 8e12, "digest": digest + tail}), flush=True)
        del out
    del b
'''
res = {}
GAMES = os.environ.get("PROBE_GAMES", "")
for form in [int(x) for x in os.environ.get("PROBE_FORMS", "0,1,2").split(",")]:
    env = dict(os.environ, OSG_ROOT=ROOT, OSG_OBS_FORM=str(form))
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=900)
    if r.returncode != 0:
        print(f"form {form} failed:", r.stderr[-2000:]); continue
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            rec = json.loads(ln)
            res.setdefault((rec["game"], rec["which"]), {})[form] = rec
for key, forms in res.items():
    base = forms.get(0)
    row = f"{key[0]:22s} which={key[1]} {str(base['shape']):18s}"
    for form, rec in sorted(forms.items()):
        same = "" if base is None or rec["digest"] == base["digest"] else "  OUTPUT DIFFERS"
        row += f" | form {form}: {rec['us']:8.1f} us {rec['frac_of_8TBs']:.3f}{same}"
    print(row)
