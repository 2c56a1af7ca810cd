"""One hex(9) wave-layout search of 8192 roots x 1024 simulations (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_spiel_amd import _abi
if os.environ.get("OSG_VARIANT_LIB"): _abi.LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"])
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
roots = osa.StateBatch(ctx, "hex(board_size=9)", 8192)
roots.mcts_search(uct_c=2.0, max_simulations=1024, seed=3, layout=2)
torch.cuda.synchronize()
