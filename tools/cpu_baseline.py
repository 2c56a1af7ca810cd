#!/usr/bin/env python3
"""The reference's CPU path beside the GPU numbers: SURVEY.md 8(d) "CPU baseline" table.

Times, on this host's cores, the loops the GPU kernels replace — once on the GENUINE reference build
(oracle/_ref/libspiel_ref.so: the reference's own .cc files, -O3 -DNDEBUG) and once on the
restatement (oracle/liboracle.so) — single thread and all usable cores (one Game object and one RNG /
solver per thread; the reference itself is single-threaded):

  env steps      Clone + LegalActions + ApplyAction + IsTerminal + Returns + CurrentPlayer per step
  tensor pack    State::ObservationTensor(player) into a preallocated span, tensors/s
  playouts       benchmark_game.cc-style random playouts, moves/s
  MCTS           MCTSBot(RandomRolloutEvaluator(1), uct_c=2, 1024 simulations), simulations/s
  CFR            CFRSolver::EvaluateAndUpdatePolicy, iterations/s
  ES-MCCFR       ExternalSamplingMCCFRSolver::RunIteration, trajectories/s (2 per iteration)

    python tools/cpu_baseline.py [--seconds 3]     # test infrastructure only; no GPU involved
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def rate(fn, seconds):
    """Calibrate with a short run, then one run sized for `seconds`."""
    units, secs = fn(1)
    scale = max(1.0, seconds / max(secs, 1e-6))
    units, secs = fn(scale)
    if secs < 0.5 * seconds:  # the short calibration run under-estimated the rate (thread start-up, cold caches)
        scale *= seconds / max(secs, 1e-6)
        units, secs = fn(scale)
    return units / secs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    args = ap.parse_args()
    import bench
    import oracle_py
    oracle_py.build()
    impls = [("restatement", oracle_py)]
    import reference_py
    if reference_py.sources_present():
        reference_py.build()
    if reference_py.available():
        impls.insert(0, ("genuine reference", reference_py))
    threads = bench.host_threads()
    print(f"host threads usable: {threads}")
    rows = []
    for name, impl in impls:
        for game in ("connect_four", "tic_tac_toe", "hex(board_size=9)", "kuhn_poker", "leduc_poker"):
            g = impl.Game(game)
            for th in (1, threads):
                def steps(scale, g=g, th=th):
                    secs, units = g.bench_env_steps(0x5EED, 1 << 12, int(50_000 * th * scale), th)
                    return units, secs
                def playouts(scale, g=g, th=th):
                    secs, moves = g.bench_playouts(0x5EED, int(500 * th * scale), th)
                    return moves, secs
                def tensors(scale, g=g, th=th):
                    secs, units = g.bench_observation(0x5EED, 1 << 12, int(200_000 * th * scale), th)
                    return units, secs
                rows.append((name, game, th, "env-steps/s", rate(steps, args.seconds)))
                rows.append((name, game, th, "ObservationTensor/s", rate(tensors, args.seconds)))
                rows.append((name, game, th, "playout moves/s", rate(playouts, args.seconds)))
        g = impl.Game("hex(board_size=9)")
        for th in (1, threads):
            def mcts(scale, g=g, th=th):
                secs, sims = g.bench_mcts(0x5EED, max(th, int(2 * th * scale)), 40, 1024, 1, 2.0, th)
                return sims, secs
            rows.append((name, "hex(board_size=9)", th, "MCTS sims/s", rate(mcts, args.seconds)))
        for game, kind, label, per_iter in (("kuhn_poker", 0, "CFR iterations/s", 1), ("leduc_poker", 0, "CFR iterations/s", 1),
                                            ("kuhn_poker", 2, "ES-MCCFR trajectories/s", 2),
                                            ("leduc_poker", 2, "ES-MCCFR trajectories/s", 2)):
            g = impl.Game(game)
            def solve(scale, g=g, kind=kind, per_iter=per_iter, game=game):
                iters = max(1, int((2000 if game == "kuhn_poker" or kind == 2 else 4) * scale))
                return iters * per_iter, g.bench_cfr(kind, iters, 1)
            rows.append((name, game, 1, label, rate(solve, args.seconds)))
    print(f"{'implementation':18s} {'game':20s} {'threads':>7s} {'metric':26s} {'rate':>12s}")
    for name, game, th, metric, value in rows:
        print(f"{name:18s} {game:20s} {th:7d} {metric:26s} {value:12.4g}")


if __name__ == "__main__":
    main()
