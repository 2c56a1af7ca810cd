import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
from open_spiel_amd import _abi
from open_spiel_amd._abi import check, lib
ctx = osa.Context(0)
n, sims = 65536, 40
roots = osa.StateBatch(ctx, "connect_four", n); roots.random_steps(3, 6)
prior = torch.full((n, 7), 1 / 7, dtype=torch.float64, device="cuda"); value = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
request = torch.zeros(n, dtype=torch.uint8, device="cuda")
cfg = _abi.MctsCfg(1.4, sims, 1, 0, 0, 0, 0, 1, 1)
tree = C.c_void_p(); check(lib().osg_mcts_tree_create(roots._h, C.byref(cfg), 9, C.byref(tree)))
leaf = osa.StateBatch(ctx, "connect_four", n)
stats = torch.empty((n, 4), dtype=torch.float64, device="cuda")
vis = torch.empty((n, 7), dtype=torch.int32, device="cuda")
for k in range(sims + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().osg_mcts_tree_advance(tree, leaf._h, prior.data_ptr(), value.data_ptr(), request.data_ptr(), 1 << 30, None))
    e1.record(); torch.cuda.synchronize()
    check(lib().osg_mcts_tree_results(tree, None, vis.data_ptr(), None, None, None, stats.data_ptr()))
    torch.cuda.synchronize()
    s = stats[:, 3]
    print(f"launch {k:3d}: {e0.elapsed_time(e1) * 1e3:7.1f} us  sims done min/mean/max {int(s.min())}/{float(s.mean()):.2f}/{int(s.max())}  nodes mean {float(stats[:, 1].mean()):.1f}  "
          f"requests {torch.bincount(request.to(torch.int64), minlength=6).tolist()}  root child visits row0 {vis[0].tolist()}", flush=True)
