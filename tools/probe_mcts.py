import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_spiel_amd import _abi
if os.environ.get("OSG_VARIANT_LIB"): _abi.LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"])
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
def roots_for(game, n, depth_mod, seed=0x5EED):
    b = osa.StateBatch(ctx, game, n)
    # advance by hash(i) mod depth_mod random moves using the device random stepper, one ply at a time
    idx = torch.arange(n, device="cuda", dtype=torch.int64)
    h = idx * 2654435761 + seed; h = h ^ (h >> 15)
    depth = ((h >> 3) % depth_mod).to(torch.int32)
    for t in range(depth_mod):
        m = b.legal_actions_mask().to(torch.float32)
        m[m.sum(1) == 0, 0] = 1.0
        a = torch.multinomial(m, 1).squeeze(1).to(torch.int32)
        a = torch.where(depth > t, a, torch.full_like(a, -1))
        trial = b.clone(); trial.apply_actions(a)
        a = torch.where(trial.is_terminal(), torch.full_like(a, -1), a)
        b.apply_actions(a)
    return b
cases = []
for layout in (1, 2):
    cases += [("hex(board_size=9)", 8192, 1024, 0, layout), ("hex(board_size=9)", 65536, 1024, 0, layout),
              ("connect_four", 65536, 256, 0, layout), ("tic_tac_toe", 65536, 1000, 0, layout)]
if len(sys.argv) > 1:
    cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0,
              int(sys.argv[5]) if len(sys.argv) > 5 else 0)]
for game, n, sims, max_nodes, layout in cases:
    b = roots_for(game, n, 40 if "hex" in game else (20 if "connect" in game else 4))
    b.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=2, max_nodes=max_nodes, layout=layout)  # sizes the pool
    torch.cuda.synchronize(); t = time.time()
    r = b.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=1, max_nodes=max_nodes, layout=layout)
    torch.cuda.synchronize(); dt = time.time() - t
    st = r["root_stats"]
    print(f"layout={layout} {game} roots={n} sims={sims}: {dt:.3f} s, {n*st[:,3].mean().item()/dt:.3e} sims/s, nodes/root mean {st[:,1].mean().item():.0f} max {st[:,1].max().item():.0f}, sims done mean {st[:,3].mean().item():.0f}", flush=True)
    del b, r
