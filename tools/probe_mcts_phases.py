"""Cycle shares of the phases of k_mcts_wave (needs a -DOSG_PHASE_TIMING build of libosg_hip.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, open_spiel_amd as osa
from open_spiel_amd import _abi
if os.environ.get("OSG_VARIANT_LIB"): _abi.LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"])  # e.g. tools/variants/libosg_pt.so
ctx = osa.Context(0)
lib = _abi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
game = sys.argv[2] if len(sys.argv) > 2 else "hex(board_size=9)"
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
roots = osa.StateBatch(ctx, game, n)
if len(sys.argv) > 4: roots.random_steps(7, int(sys.argv[4]))    # a position that many random plies into the game
roots.mcts_search(uct_c=2.0, max_simulations=sims, seed=3, layout=2)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
lib.osg_debug_phase_cycles(out, 1)
t = time.time(); roots.mcts_search(uct_c=2.0, max_simulations=sims, seed=4, layout=2); torch.cuda.synchronize(); dt = time.time() - t
lib.osg_debug_phase_cycles(out, 0)
names = ["loop head (terminal/legal)", "expand", "select (scan/UCT/argmax)", "apply", "playout: key threshold", "playout: fill + flood", "backup+solver", "sim setup"]
tot = sum(out[i] for i in range(8))
print(f"{game} {n} roots x {sims} simulations: {dt:.4f} s = {n * sims / dt:.3e} simulations/s")
for i, nm in enumerate(names):
    print(f"{nm:32s} {out[i] / (n * sims):9.1f} cycles/sim  {100.0 * out[i] / tot:5.1f} %")
