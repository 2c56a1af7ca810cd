"""One launch of the subtree CFR kernel on 3-player leduc_poker (100 iterations) for counter passes
(tools/pmc_kernels.sh <tag> k_cfr_sub tools/probe_cfr_sub_once.py OSG_CFR_PLAIN_LAUNCH=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import open_spiel_amd as osa
ctx = osa.Context(0)
s = osa.TabularSolver(ctx, "leduc_poker(players=3)")
s.evaluate_and_update_policy(2); ctx.synchronize()
s.evaluate_and_update_policy(100); ctx.synchronize()
print(s.last_kernel(), s.iteration)
