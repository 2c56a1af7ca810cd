"""leduc_poker ES-MCCFR by QUALITY per second: time (traversal + fold launches only; NashConv evaluations are
not timed) until the average policy's NashConv drops below 1.0 / 0.3 / 0.1 / 0.05, per mini-batch size.
The CPU line beside it: the genuine reference's sequential RunIteration (oracle/_ref when present)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch, open_spiel_amd as osa

THRESHOLDS = (1.0, 0.3, 0.1, 0.05)


def gpu_curve(ctx, game, batch, budget_s=2.5, seed=0x5EED):
    s = osa.TabularSolver(ctx, game, mccfr=True)
    s.run_mccfr(seed, 64)  # warm-up launch (allocations), then start over
    s.reset()
    torch.cuda.synchronize()
    reached, spent, updates, first, check = {}, 0.0, 0, 0, 1
    while spent < budget_s and len(reached) < len(THRESHOLDS):
        todo = check - updates
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(todo):
            s.run_mccfr(seed, batch, first_trajectory=first)
            first += batch
        torch.cuda.synchronize(); spent += time.perf_counter() - t0
        updates = check
        nc = s.nash_conv()
        for th in THRESHOLDS:
            if nc <= th and th not in reached:
                reached[th] = (spent, updates)
        check = max(check + 1, int(check * 1.3))
    return reached, spent, updates, s.nash_conv()


def main():
    ctx = osa.Context(0)
    game = sys.argv[1] if len(sys.argv) > 1 else "leduc_poker"
    for lb in (8, 10, 12, 14, 16, 18, 20):
        reached, spent, updates, nc = gpu_curve(ctx, game, 1 << lb)
        cells = "  ".join(f"<= {th}: " + (f"{reached[th][0] * 1e3:8.2f} ms ({reached[th][1]} upd)" if th in reached else "   --   ")
                          for th in THRESHOLDS)
        print(f"GPU {game} batch 2^{lb}: {cells}   [{updates} updates in {spent:.2f} s, {updates * (1 << lb) / spent:.3g} traj/s, NashConv {nc:.4f}]",
              flush=True)
    # full-tree CFR on the device for comparison (exact updates, no sampling)
    s = osa.TabularSolver(ctx, game)
    s.evaluate_and_update_policy(1); s.reset(); torch.cuda.synchronize()
    reached, spent, it, step = {}, 0.0, 0, 1
    while spent < 2.5 and len(reached) < len(THRESHOLDS):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.evaluate_and_update_policy(step)
        torch.cuda.synchronize(); spent += time.perf_counter() - t0
        it += step
        nc = s.nash_conv()
        for th in THRESHOLDS:
            if nc <= th and th not in reached:
                reached[th] = (spent, it)
        step = max(1, it // 3)
    print(f"GPU {game} CFRSolver (full tree): " + "  ".join(f"<= {th}: {reached[th][0] * 1e3:8.2f} ms ({reached[th][1]} it)" if th in reached else "--" for th in THRESHOLDS), flush=True)
    try:
        import reference_py as impl
        kind = "reference" if impl.available() else None
    except Exception:
        kind = None
    if kind is None:
        import oracle_py as impl
        impl.build(); kind = "port"
    sol = impl.Solver(impl.Game(game), "mccfr_simple", 0x5EED)
    reached, spent, it, step = {}, 0.0, 0, 100
    while spent < 25.0 and len(reached) < len(THRESHOLDS):
        t0 = time.perf_counter(); sol.iterate(step); spent += time.perf_counter() - t0
        it += step
        nc = sol.nash_conv()
        for th in THRESHOLDS:
            if nc <= th and th not in reached:
                reached[th] = (spent, it)
        step = max(100, it // 3)
    print(f"CPU {kind} {game} ExternalSamplingMCCFRSolver (1 thread, sequential by construction): " +
          "  ".join(f"<= {th}: {reached[th][0]:7.2f} s ({reached[th][1]} it)" if th in reached else f"<= {th}: --" for th in THRESHOLDS) +
          f"   [{it} iterations in {spent:.1f} s, NashConv {sol.nash_conv():.4f}]", flush=True)


if __name__ == "__main__":
    main()
