"""Network-guided search throughput (open_spiel_amd/mcts.py): connect_four, a small torch MLP as the
value / policy network (random weights), PUCT with root noise, one forward per evaluator round."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_spiel_amd import _abi
if os.environ.get("OSG_VARIANT_LIB"): _abi.LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"])
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

def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

# what one round of 65536 leaves costs, piece by piece (eager launches, so each figure includes its launch gaps)
big = osa.StateBatch(ctx, "connect_four", 65536); big.random_steps(3, 9)
ev = mcts.VPNetEvaluator(model)
obs = big.observation_tensor(-1)
legal = big.legal_actions_mask()[:, :7].bool()
every = torch.ones(65536, dtype=torch.bool, device="cuda")
print(f"65536 leaves: observation pack {timeit(lambda: big.observation_tensor(-1, out=obs)) * 1e6:.1f} us, legal mask (bits -> [n, 7] bool) "
      f"{timeit(lambda: big.legal_actions_bool()) * 1e6:.1f} us, network forward {timeit(lambda: net(obs)) * 1e6:.1f} us, "
      f"whole model (forward + masked softmax + tanh) {timeit(lambda: model(obs, legal)) * 1e6:.1f} us, "
      f"evaluator.evaluate {timeit(lambda: ev.evaluate(big, every, every)) * 1e6:.1f} us", flush=True)
del big, obs, legal

for n, sims in [(1024, 200), (4096, 200), (16384, 200), (65536, 100)]:
    roots = osa.StateBatch(ctx, "connect_four", n)
    roots.random_steps(3, 6)
    mcts.search(roots, mcts.VPNetEvaluator(model), max_simulations=8, puct=True)  # warm-up
    calls[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = mcts.search(roots, mcts.VPNetEvaluator(model), max_simulations=sims, uct_c=1.4, puct=True, dirichlet_alpha=0.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    done = float(res["root_stats"][:, 3].sum())
    print(f"connect_four {n} roots x {sims} sims, MLP 126-256-256-8 fp32, one evaluator round per simulation (prior kept with the value, nothing read back between rounds): {dt:.3f} s, {done / dt:.3g} sims/s "
          f"({dt / (sims + 1) * 1e3:.3f} ms per round: search kernel + observation pack + legal mask + forward)", flush=True)
    calls[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res0 = mcts.search(roots, mcts.VPNetEvaluator(model), max_simulations=sims, uct_c=1.4, puct=True, graph=False)
    torch.cuda.synchronize(); dt0 = time.perf_counter() - t0
    same = bool(torch.equal(res["child_visits"], res0["child_visits"]))
    print(f"   request / answer loop (round 2's path): {float(res0['root_stats'][:, 3].sum()) / dt0:.3g} sims/s, {calls[0]} forwards, "
          f"{dt0 / calls[0] * 1e3:.2f} ms per round; same visit counts: {same}", flush=True)
    ro = mcts.search(roots, mcts.RolloutEvaluator(), max_simulations=sims, uct_c=1.4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ro = mcts.search(roots, mcts.RolloutEvaluator(), max_simulations=sims, uct_c=1.4)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    fused = roots.mcts_search(uct_c=1.4, max_simulations=sims, layout=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fused = roots.mcts_search(uct_c=1.4, max_simulations=sims, layout=1)
    torch.cuda.synchronize(); dt3 = time.perf_counter() - t0
    print(f"   same roots, rollout evaluator outside the kernel: {done / dt2:.3g} sims/s; fused kernel: {float(fused['root_stats'][:, 3].sum()) / dt3:.3g} sims/s", flush=True)
