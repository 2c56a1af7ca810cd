"""The two solver kernels bench.py times for configs 3 and 5, a few launches each, for counter passes
(tools/pmc_solvers.sh): kuhn_poker CFR, 3 launches of 1 000 iterations (k_cfr_small<lds, owner>), and leduc_poker
ES-MCCFR, 4 mini-batches of 2^20 trajectories (k_mccfr_resident_flat<3>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import open_spiel_amd as osa
ctx = osa.Context(0)
c = osa.TabularSolver(ctx, "kuhn_poker")
for _ in range(3):
    c.evaluate_and_update_policy(1000)
ctx.synchronize()
assert c.last_kernel() == "k_cfr_small<lds, owner>"
s = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
for k in range(4):
    s.run_mccfr(0x5EED, 1 << 20, first_trajectory=k << 20)
ctx.synchronize()
assert s.last_kernel().startswith("k_mccfr_resident_flat"), s.last_kernel()
print("mccfr kernel form:", s.last_kernel())
