"""leduc_poker ES-MCCFR, 40 mini-batches of 2^14 trajectories (for counter passes: tools/pmc_kernels.sh <tag> k_mccfr_resident ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import open_spiel_amd as osa
ctx = osa.Context(0)
s = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
for k in range(40):
    s.run_mccfr(7, 1 << 14, first_trajectory=k << 14)
ctx.synchronize()
