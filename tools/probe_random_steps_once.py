"""The bench's K = 32 persistent leg, alone: connect_four, 2^20 states, 32 random steps per launch (for counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, open_spiel_amd as osa
ctx = osa.Context(0)
n = 1 << 20
b = osa.StateBatch(ctx, "connect_four", n)
c = torch.zeros(2, dtype=torch.int64, device="cuda")
for j in range(3):
    b.random_steps(9 + j, 32, counters=c)
torch.cuda.synchronize()
t = time.time()
for j in range(10):
    b.random_steps(20 + j, 32, counters=c)
torch.cuda.synchronize()
dt = (time.time() - t) / 10
print(f"k=32 {dt*1e6:.1f} us/launch {n*32/dt:.3e} steps/s", flush=True)
