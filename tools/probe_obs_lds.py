"""tic_tac_toe / leduc_poker tensors at 2^24 states: the piece kernels with the rows' images staged in LDS (OSG_OBS_LDS=4 / 8)
against every piece building its own (OSG_OBS_LDS=0).  One process per setting (the library reads the variable once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
for game, which in (("tic_tac_toe", 0), ("leduc_poker", 0), ("leduc_poker", 1), ("kuhn_poker", 0)):
    n = 1 << 24
    b = osa.StateBatch(ctx, game, n); b.random_steps(3, 4)
    size = b.desc.obs_size if which == 0 else b.desc.info_size
    out = torch.empty((n, size), dtype=torch.float32, device="cuda")
    f = (lambda: b.observation_tensor(0, out=out)) if which == 0 else (lambda: b.information_state_tensor(0, out=out))
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 10 * 1e6
    bytes_ = n * (size * 4 + b.desc.state_words * (4 if game == "tic_tac_toe" else 8))
    print(f"{game:12s} which={which} [{n}, {size}]  {us:8.1f} us  {bytes_ / us / 8e6:.3f} of 8 TB/s  checksum {float(out[::4097].sum()):.1f}", flush=True)
    del out, b
''' % ROOT
for lds in ("default", "0", "4", "8"):
    print(f"-- OSG_OBS_LDS={lds}", flush=True)
    env = dict(os.environ)
    env.pop("OSG_OBS_LDS", None)
    if lds != "default":
        env["OSG_OBS_LDS"] = lds
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
