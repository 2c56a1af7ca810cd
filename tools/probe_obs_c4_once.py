import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
game = os.environ.get("PROBE_GAME", "connect_four")
n = int(os.environ.get("PROBE_N", 1 << 24))
b = osa.StateBatch(ctx, game, n); b.random_steps(3, int(os.environ.get("PROBE_DEPTH", 12)))
which = int(os.environ.get("PROBE_WHICH", 0))
size = b.desc.obs_size if which == 0 else b.desc.info_size
out = torch.empty((n, size), dtype=torch.float32, device="cuda")
for _ in range(4):
    if which == 0: b.observation_tensor(0, out=out)
    else: b.information_state_tensor(0, out=out)
torch.cuda.synchronize()
