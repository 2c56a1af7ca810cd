"""BASELINE.json configs[0] on its own: bench.ttt_mcts_config1 (device single-root MCTSBot beside the CPU reference)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.ttt_mcts_config1(True), indent=1))
