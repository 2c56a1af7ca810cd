import os, sys
sys.path.insert(0, os.getcwd())
import time, torch, open_spiel_amd as osa
ctx = osa.Context(0)
for replicas in (4096, 8192, 16384, 32768):
    s = osa.TabularSolver(ctx, "kuhn_poker", replicas=replicas, random_initial_regrets=True, seed=7)
    s.evaluate_and_update_policy(10); torch.cuda.synchronize()
    t = time.time(); s.evaluate_and_update_policy(2000); torch.cuda.synchronize(); dt = time.time() - t
    print(f"replicas {replicas}: {replicas*2000/dt:.3e} solver-iterations/s ({dt*1e3:.1f} ms)", flush=True)
