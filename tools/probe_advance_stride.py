"""k_mcts_advance alone (65536 connect_four roots, constant evaluator answers): kernel time per launch by HIP events,
for lane strides 1 / 2 / 4 / 8 / 16 (OSG_MCTS_LANE_STRIDE)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
from open_spiel_amd import _abi
from open_spiel_amd._abi import check, lib
ctx = osa.Context(0)
n, sims = 65536, 100
roots = osa.StateBatch(ctx, "connect_four", n); roots.random_steps(3, 6)
prior = torch.full((n, 7), 1 / 7, dtype=torch.float64, device="cuda"); value = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
request = torch.zeros(n, dtype=torch.uint8, device="cuda")
for flags in ((1 | 8,) if os.environ.get('OSG_PROBE_ONE') else (1 | 8, 1)):
  for stride in ((1,) if os.environ.get('OSG_PROBE_ONE') else (1, 1)):
    os.environ["OSG_MCTS_LANE_STRIDE"] = str(stride)
    cfg = _abi.MctsCfg(1.4, sims, 1, 0, 0, 0, 0, 1, 1)
    tree = C.c_void_p(); check(lib().osg_mcts_tree_create(roots._h, C.byref(cfg), flags, C.byref(tree)))
    leaf = osa.StateBatch(ctx, "connect_four", n)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(sims + 2)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rounds = sims + 1 if flags & 8 else 2 * sims + 2
    for k in range(sims + 1):
        e[k].record()
        check(lib().osg_mcts_tree_advance(tree, leaf._h, prior.data_ptr(), value.data_ptr(), request.data_ptr(), 1 << 30, None))
    e[sims + 1].record(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    per = [e[k].elapsed_time(e[k + 1]) * 1e3 for k in range(sims + 1)]
    print(f"flags {flags} lane stride {stride:2d}: {sum(per) / len(per):7.1f} us per launch (first 10: {sum(per[:10]) / 10:6.1f}, last 10: {sum(per[-10:]) / 10:6.1f}); wall {wall * 1e3:.1f} ms", flush=True)
    print("   per launch (us):", " ".join(f"{x:.0f}" for x in per), flush=True)
    lib().osg_mcts_tree_destroy(tree); del leaf
