"""Where a round of the network-guided search goes (65536 connect_four roots): captured graph vs eager launches, with
the real evaluator and with an evaluator that returns constants (= the search kernel + the two answer copies alone)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
from open_spiel_amd import mcts
ctx = osa.Context(0)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(126, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Linear(256, 8)).cuda()
def model(obs, legal):
    out = net(obs)
    return torch.softmax(out[:, :7].masked_fill(~legal, -1e9), 1), torch.tanh(out[:, 7])
class Const(mcts.BatchedEvaluator):
    joint = True
    def __init__(self, n):
        self.p = torch.full((n, 7), 1 / 7, dtype=torch.float64, device="cuda"); self.v = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
    def evaluate(self, leaf, wp, wv): return self.p, self.v
n, sims = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 100
roots = osa.StateBatch(ctx, "connect_four", n); roots.random_steps(3, 6)
def bf16_model(obs, legal):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(obs).float()
    return torch.softmax(out[:, :7].masked_fill(~legal, -1e9), 1), torch.tanh(out[:, 7])
cases = [("eager, network fp32", mcts.VPNetEvaluator(model), None, None), ("eager, network bf16 autocast", mcts.VPNetEvaluator(bf16_model), None, None)]
cases += [(f"eager, constant answers, lane stride {k}", Const(n), None, k) for k in (1, 2, 4, 8, 16)]
cases += [("graph, network fp32", mcts.VPNetEvaluator(model), True, None)]
for label, ev, graph, stride in cases:
    if stride: os.environ["OSG_MCTS_LANE_STRIDE"] = str(stride)
    else: os.environ.pop("OSG_MCTS_LANE_STRIDE", None)
    if graph is None:
        import open_spiel_amd.mcts as M
        run = lambda: M._search_joint(roots, ev, sims, 1.4, 1, False, 0, 0, 0, True, False, None, use_graph=False)
    else:
        run = lambda: mcts.search(roots, ev, max_simulations=sims, uct_c=1.4, puct=True, graph=True)
    run(); torch.cuda.synchronize(); t0 = time.perf_counter(); res = run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n} roots x {sims} sims [{label}]: {dt * 1e3:.1f} ms, {float(res['root_stats'][:, 3].sum()) / dt:.3g} sims/s, {dt / sims * 1e3:.3f} ms per round", flush=True)
