"""The judge (k_policy_eval: expected returns + every player's best response, one workgroup) and CFR-BR on kuhn and leduc."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
for game in ("kuhn_poker", "leduc_poker"):
    s = osa.TabularSolver(ctx, game)
    s.evaluate_and_update_policy(100); torch.cuda.synchronize()
    t = time.time()
    for _ in range(200): s.nash_conv()
    dt = (time.time() - t) / 200
    print(f"{game}: nash_conv() {dt * 1e6:.1f} us per call (host round trip included)", flush=True)
    b = osa.TabularSolver(ctx, game)
    b.evaluate_and_update_policy_cfr_br(5); torch.cuda.synchronize()
    t = time.time(); b.evaluate_and_update_policy_cfr_br(300); torch.cuda.synchronize(); dt = (time.time() - t) / 300
    print(f"{game}: CFR-BR {1 / dt:.1f} iterations/s ({dt * 1e6:.1f} us)", flush=True)
