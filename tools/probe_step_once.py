"""One fused-step workload for the counter passes (tools/pmc_kernels.sh): PROBE_GAME at PROBE_N states, uniformly random
legal actions, out of place, 6 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
game = os.environ.get("PROBE_GAME", "hex(board_size=9)")
n = int(os.environ.get("PROBE_N", 1 << 24))
b = osa.StateBatch(ctx, game, n); b.random_steps(3, int(os.environ.get("PROBE_DEPTH", 30)))
lm = b.legal_actions_mask()
acts = torch.where(lm.any(1), (lm.to(torch.float32) * torch.rand(lm.shape, device="cuda")).argmax(1),
                   torch.full((n,), 255, device="cuda")).to(torch.uint8)
del lm
dst = osa.StateBatch(ctx, game, n)
mask, status = b.step_buffers()
for _ in range(6):
    b.step(acts, dst=dst, mask=mask, status=status)
torch.cuda.synchronize()
