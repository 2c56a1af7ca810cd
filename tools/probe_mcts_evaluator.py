"""Network-guided search throughput (open_spiel_amd/mcts.py): connect_four, a small torch MLP as the
value / policy network (random weights), PUCT with root noise, one forward per evaluator round."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
from open_spiel_amd import mcts

ctx = osa.Context(0)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(126, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(),
                          torch.nn.Linear(256, 8)).cuda()
calls = [0]

def model(obs, legal):
    calls[0] += 1
    out = net(obs)
    policy = torch.softmax(out[:, :7].masked_fill(~legal, -1e9), 1)
    return policy, torch.tanh(out[:, 7])

for n, sims in [(1024, 200), (4096, 200), (16384, 200), (65536, 100)]:
    roots = osa.StateBatch(ctx, "connect_four", n)
    roots.random_steps(3, 6)
    mcts.search(roots, mcts.VPNetEvaluator(model), max_simulations=8, puct=True)  # warm-up
    calls[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = mcts.search(roots, mcts.VPNetEvaluator(model), max_simulations=sims, uct_c=1.4, puct=True, dirichlet_alpha=0.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    done = float(res["root_stats"][:, 3].sum())
    print(f"connect_four {n} roots x {sims} sims, MLP 126-256-256-8: {dt:.3f} s, {done / dt:.3g} sims/s, {calls[0]} forwards "
          f"({dt / calls[0] * 1e3:.2f} ms per round incl. the search kernel, observation pack and legal mask)", flush=True)
    ro = mcts.search(roots, mcts.RolloutEvaluator(), max_simulations=sims, uct_c=1.4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ro = mcts.search(roots, mcts.RolloutEvaluator(), max_simulations=sims, uct_c=1.4)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    fused = roots.mcts_search(uct_c=1.4, max_simulations=sims, layout=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fused = roots.mcts_search(uct_c=1.4, max_simulations=sims, layout=1)
    torch.cuda.synchronize(); dt3 = time.perf_counter() - t0
    print(f"   same roots, rollout evaluator outside the kernel: {done / dt2:.3g} sims/s; fused kernel: {float(fused['root_stats'][:, 3].sum()) / dt3:.3g} sims/s", flush=True)
