import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, open_spiel_amd as osa
ctx = osa.Context(0)
for game in ("connect_four", "tic_tac_toe"):
    for n in (1 << 20, 1 << 17):
        b = osa.StateBatch(ctx, game, n)
        c = torch.zeros(2, dtype=torch.int64, device="cuda")
        for steps in (1, 32, 320):
            b.random_steps(9, steps, counters=c); torch.cuda.synchronize()
            t = time.time()
            for _ in range(10): b.random_steps(9, steps, counters=c)
            torch.cuda.synchronize(); dt = (time.time() - t) / 10
            print(game, n, steps, f"{dt*1e6:.1f} us", f"{n*steps/dt:.3e} steps/s", flush=True)
