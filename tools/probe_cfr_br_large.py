"""CFR-BR on 3-player leduc_poker (1.83 M histories): the persistent pass set (k_cfr_sub<., kBr>, one launch per iteration
after the evaluation's sweep) against a launch per phase (k_gcfr<br>); tables compared bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
tabs = {}
for form in (False, "grid", False, "grid"):
    s = osa.TabularSolver(ctx, "leduc_poker(players=3)", general_kernel=form)
    s.evaluate_and_update_policy(2)
    s.evaluate_and_update_policy_cfr_br(3); ctx.synchronize()
    t0 = time.perf_counter(); s.evaluate_and_update_policy_cfr_br(100); ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"leduc_poker(players=3) cfr-br [{s.last_kernel()}]: {100 / dt:.0f} it/s ({dt / 100 * 1e3:.3f} ms per iteration)", flush=True)
    tabs[form] = s.tables()
    t0 = time.perf_counter(); nc = s.nash_conv(); dt = time.perf_counter() - t0
    print(f"   nash_conv {nc:.6f} ({dt * 1e3:.2f} ms)", flush=True)
    del s
for name in ("regrets", "cum_policy", "cur_policy"):
    print(name, "bit-identical:", bool(np.array_equal(tabs[False][name], tabs["grid"][name])))

# the evaluation alone: a launch per level and phase (default) against one persistent launch (OSG_EVAL_PERSIST=1)
for form in ("0", "1", "0", "1"):
    os.environ["OSG_EVAL_PERSIST"] = form
    s2 = osa.TabularSolver(ctx, "leduc_poker(players=3)")
    s2.evaluate_and_update_policy(3)
    s2.nash_conv(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): nc = s2.nash_conv()
    dt = (time.perf_counter() - t0) / 50
    print(f"nash_conv [{s2.last_eval_kernel()}]: {dt * 1e3:.3f} ms per call ({nc:.9f})", flush=True)
    t0 = time.perf_counter(); s2.evaluate_and_update_policy_cfr_br(50); ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"   cfr-br [{s2.last_kernel()} + {s2.last_eval_kernel()}]: {50 / dt:.0f} it/s", flush=True)
    del s2
os.environ.pop("OSG_EVAL_PERSIST", None)
