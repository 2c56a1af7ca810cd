import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np, open_spiel_amd as osa
from open_spiel_amd import mcts
from test_z5_gpu_mcts_evaluator import StubNet, StubEvaluator, JointStubEvaluator
ctx = osa.Context(0)
roots = osa.StateBatch(ctx, "connect_four", 8)
roots.random_steps(3, 5)
net = StubNet(roots.desc.obs_size, 7, 2, ctx.device)
for max_nodes in (0, 60):
    first_bad = None
    for sims in list(range(2, 80)) + [100, 150, 200, 300, 400]:
        a = mcts.search(roots, StubEvaluator(net), max_simulations=sims, uct_c=1.3, puct=True, seed=7, max_nodes=max_nodes, graph=False, want_tree_of=0)
        b = mcts.search(roots, JointStubEvaluator(net), max_simulations=sims, uct_c=1.3, puct=True, seed=7, max_nodes=max_nodes, graph=None, want_tree_of=0)
        same = torch.equal(a["child_visits"], b["child_visits"]) and torch.equal(a["child_reward"], b["child_reward"])
        if not same and first_bad is None:
            first_bad = sims
            bad_roots = (a["child_visits"] != b["child_visits"]).any(1).nonzero().flatten().tolist()
            print("max_nodes", max_nodes, "first difference at sims", sims, "roots", bad_roots)
            print(" loop  visits", a["child_visits"][bad_roots[0]].tolist(), "stats", a["root_stats"][bad_roots[0]].tolist())
            print(" joint visits", b["child_visits"][bad_roots[0]].tolist(), "stats", b["root_stats"][bad_roots[0]].tolist())
            if bad_roots[0] == 0:
                for name, t in (("loop", a["tree"]), ("joint", b["tree"])):
                    print(name, "nodes", len(t["meta"]))
                    for i in range(min(len(t["meta"]), 40)):
                        m = int(t["meta"][i])
                        print(f"   {i:3d} act {m & 0xFF:3d} pl {((m >> 8) & 15) - 1} nch {(m >> 12) & 0xFF} first {int(t['first_child'][i]):4d} cnt {int(t['explore_count'][i]):3d} tot {t['total_reward'][i]:+.4f} prior {t['prior'][i]:.4f}")
            break
    print("max_nodes", max_nodes, "first_bad", first_bad)
