"""Where one single-root search's time goes (tic_tac_toe, 1000 simulations x 20 playouts, solve): tree creation, the ONE
launch that runs the whole search (two wavefronts), the results, the tree download — through the C-ABI, host timers
around synchronised calls."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, open_spiel_amd as osa
from open_spiel_amd import _abi
from open_spiel_amd._abi import check, lib
ctx = osa.Context(0)
game = os.environ.get("PROBE_GAME", "tic_tac_toe")
sims = int(os.environ.get("PROBE_SIMS", 1000))
roots = osa.StateBatch(ctx, game, 1)
A = roots.num_distinct_actions
leaf = osa.StateBatch(ctx, game, 1)
request = torch.zeros(1, dtype=torch.uint8, device="cuda")
rows = []
for rep in range(6):
    cfg = _abi.MctsCfg(2.0, sims, 20, 1, 0, 42, 0, 1, 0)
    tree = C.c_void_p()
    ctx.synchronize(); t0 = time.perf_counter()
    check(lib().osg_mcts_tree_create(roots._h, C.byref(cfg), 4, C.byref(tree)))
    ctx.synchronize(); t1 = time.perf_counter()
    counts = (C.c_int64 * 4)()
    check(lib().osg_mcts_tree_advance(tree, leaf._h, None, None, request.data_ptr(), sims, counts))
    ctx.synchronize(); t2 = time.perf_counter()
    best = torch.zeros(1, dtype=torch.int32, device="cuda"); visits = torch.zeros((1, A), dtype=torch.int32, device="cuda")
    reward = torch.zeros((1, A), dtype=torch.float64, device="cuda"); outcome = torch.zeros((1, A), dtype=torch.int8, device="cuda")
    stats = torch.zeros((1, 4), dtype=torch.float64, device="cuda")
    check(lib().osg_mcts_tree_results(tree, best.data_ptr(), visits.data_ptr(), reward.data_ptr(), outcome.data_ptr(), stats.data_ptr(), 0))
    ctx.synchronize(); t3 = time.perf_counter()
    nodes = lib().osg_mcts_tree_nodes(tree, 0)
    lib().osg_mcts_tree_destroy(tree)
    done = int(stats[0, 3].item())
    rows.append((t1 - t0, t2 - t1, t3 - t2, done, nodes))
for c, a, r, done, nodes in rows[1:]:
    print(f"create {c * 1e6:8.1f} us   search launch {a * 1e6:9.1f} us ({done} simulations, {a / max(done, 1) * 1e6:6.2f} us each, "
          f"{done / a:9.0f} sims/s)   results {r * 1e6:7.1f} us   nodes {nodes}")
