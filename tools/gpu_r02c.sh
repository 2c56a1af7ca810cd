set -u
OUT=gpurun_out/r02c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt; tail -5 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
# hex(9) MCTS: the default 16384-node pool vs a pool that can never run out (1 + sims * 81)
timeout 600 python - > $OUT/mcts_pool.log 2>&1 <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
import open_spiel_amd as osa, bench
ctx = osa.Context(0)
for n in (1 << 13, 1 << 16):
    roots = bench.hex_roots(osa, torch, ctx, n, 0)
    for max_nodes in (0, 1 + 1024 * 81):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = roots.mcts_search(uct_c=2.0, max_simulations=1024, n_rollouts=1, seed=bench.SEED, max_nodes=max_nodes)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = res["root_stats"]
        print(f"roots {n} max_nodes {max_nodes}: {dt:.4f} s, {float(st[:,3].sum())/dt:.4g} sims/s, nodes used mean {float(st[:,1].mean()):.0f} max {float(st[:,1].max()):.0f}", flush=True)
    del roots
PY
cat $OUT/mcts_pool.log | tee -a $OUT/summary.txt
