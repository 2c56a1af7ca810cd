import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
game = sys.argv[1] if len(sys.argv) > 1 else "hex(board_size=9)"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
b = osa.StateBatch(ctx, game, n); b.random_steps(3, 30)
out = torch.empty((n, b.desc.obs_size), dtype=torch.float32, device="cuda")
for _ in range(20): b.observation_tensor(0, out=out)
torch.cuda.synchronize()
