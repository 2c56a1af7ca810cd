"""The evaluator-outside search (k_mcts_advance, one simulation per launch) with the node pool root-major against root-minor
(OSG_STEP_ROOT_MAJOR): random-rollout evaluator outside the kernel, 2^14 roots."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
from open_spiel_amd import mcts
ctx = osa.Context(0)
for game, n, sims in (("hex(board_size=9)", 1 << 14, 128), ("hex", 1 << 14, 128), ("connect_four", 1 << 14, 200), ("tic_tac_toe", 1 << 14, 200)):
    roots = osa.StateBatch(ctx, game, n); roots.random_steps(3, 4)
    for rm in ("0", "1", "0", "1"):
        os.environ["OSG_STEP_ROOT_MAJOR"] = rm
        mcts.search(roots, mcts.RolloutEvaluator(), max_simulations=8, uct_c=1.4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = mcts.search(roots, mcts.RolloutEvaluator(), max_simulations=sims, uct_c=1.4)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{game:20s} roots {n} x {sims} sims, root_major={rm}: {float(res['root_stats'][:, 3].sum()) / dt:.3e} sims/s", flush=True)
