"""Digest of a wave-per-root search's outputs on probe_mcts.py's roots (for comparing two builds of the library)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
torch.manual_seed(1234)
import open_spiel_amd as osa
ctx = osa.Context(0)
game, n, sims = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
b = osa.StateBatch(ctx, game, n)
idx = torch.arange(n, device="cuda", dtype=torch.int64)
h = idx * 2654435761 + 0x5EED; h = h ^ (h >> 15)
depth = ((h >> 3) % 40).to(torch.int32)
for t in range(40):
    m = b.legal_actions_mask().to(torch.float32)
    m[m.sum(1) == 0, 0] = 1.0
    a = torch.multinomial(m, 1).squeeze(1).to(torch.int32)
    a = torch.where(depth > t, a, torch.full_like(a, -1))
    trial = b.clone(); trial.apply_actions(a)
    a = torch.where(trial.is_terminal(), torch.full_like(a, -1), a)
    b.apply_actions(a)
words = b.words().cpu().numpy() if hasattr(b, "words") else None
for rep in range(2):
    r = b.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=7, layout=2)
    v = r["child_visits"].cpu().numpy(); rw = r["child_reward"].cpu().numpy(); st = r["root_stats"].cpu().numpy()
    print(game, "rep", rep, "visits", hashlib.sha1(v.tobytes()).hexdigest()[:12], "reward", hashlib.sha1(rw.tobytes()).hexdigest()[:12],
          "stats", hashlib.sha1(np.nan_to_num(st, nan=-7.0).tobytes()).hexdigest()[:12], "nodes mean", st[:, 1].mean(), flush=True)
np.save(f"gpurun_out/r06zzf/visits_{os.environ.get('TAG','x')}_{game.replace('(','_').replace(')','').replace('=','')}.npy", v)
