#!/usr/bin/env python3
"""Per-launch duration of the headline kernel, untraced beside traced, from one validation pass.

untraced: bench.py's own HIP-event timing (`roofline.avg_launch_us`, and the DRAM leg's), no profiler attached;
traced  : `rocprofv3 --kernel-trace --stats` of the same command (tools/rocpd_summary.py's second table, grouped by
          grid size: the 2^20-state launches and the 2^24-state launches of the DRAM leg are separate rows).
Usage: launch_timing.py gpurun_out/<tag>  ->  <tag>/launch_timing.txt (stdout)."""
import csv, json, os, sys

out = sys.argv[1]
line = [l for l in open(os.path.join(out, "bench_n1.log")) if l.startswith("{")][-1]
j = json.loads(line)
r = j["roofline"]
kernel = r.get("kernel", "k_step")
print(f"kernel {kernel}")
print(f"untraced (HIP events inside bench.py, {r.get('launches_timed')} launches): {r['avg_launch_us']:.3f} us per launch at 2^20 states"
      f" -> {r['achieved']:.0f} GB/s algorithmic, frac {r['frac']:.3f}")
if r.get("hbm_avg_launch_us") is not None:   # (the compact line: scalars beside the headline's)
    print(f"untraced, DRAM leg ({r.get('hbm_states', '2^24')} states): {r['hbm_avg_launch_us']:.3f} us per launch, frac {r.get('hbm_frac')}")
for leg_name in ("dram_leg", "dram"):
    leg = r.get(leg_name) or j.get(leg_name)
    if isinstance(leg, dict) and "avg_launch_us" in leg:
        print(f"untraced, DRAM leg ({leg.get('states', '2^24')} states): {leg['avg_launch_us']:.3f} us per launch, frac {leg.get('frac')}")
algo = r.get("algorithmic_bytes_per_launch")
path = os.path.join(out, "kernel_stats.csv")
if os.path.exists(path):
    rows = list(csv.reader(open(path)))
    second = False
    for row in rows:
        if not row:
            second = True
            continue
        if second and row[0] != "Name" and kernel in row[0]:
            name, calls, mn, mx, avg, grid = row[0], row[1], float(row[2]), float(row[3]), float(row[4]), row[5]
            print(f"traced   (rocprofv3 --kernel-trace, grid_x {grid}, {calls} launches): avg {avg / 1e3:.3f} us, min {mn / 1e3:.3f} us, max {mx / 1e3:.3f} us")
            # the fraction on the TRACED duration (the kernel's own begin-to-end time): 35 B x states / avg / 8 TB/s, states from the grid
            try:
                states = int(grid) * (2 if "c4std2" in name else 1)   # grid_x counts threads; k_step_c4std2 steps two states per thread
                print(f"         frac_traced = 35 B x {states} states / {avg / 1e3:.3f} us / 8 TB/s = {35 * states / (avg * 1e-9) / 8e12:.3f}")
            except ValueError:
                pass
else:
    print("traced: no kernel_stats.csv in", out)
print("untraced = (event after the last launch - event before the first) / launches on a saturated stream; traced = the kernels'"
      " own begin-to-end durations.  Consecutive launches overlap head to tail on a saturated stream, so the untraced figure is the"
      " SMALLER time: bench.py's `value` and `roofline.frac` (which use it, as the contract's timed region does) are the favourable"
      " reading, and frac_traced above is the conservative one; bench_detail.json carries both where a trace exists.")
