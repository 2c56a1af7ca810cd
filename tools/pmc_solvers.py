"""Summarises the counter passes of tools/pmc_solvers.sh into profiles/<tag>_pmc_solvers.json: per kernel (k_cfr_small:
kuhn CFR, 1 000 iterations per launch; k_mccfr_resident_flat: leduc ES-MCCFR, 2^20 trajectories per launch) the mean of
every counter per launch, the launch durations under the counters, and the per-unit figures bench.py's
secondary.cfr.roofline / secondary.mccfr.roofline quote.   python tools/pmc_solvers.py gpurun_out/<tag>/pmc_solvers <tag>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir, tag = sys.argv[1], sys.argv[2]
KERNELS = {"k_cfr_small": ("iterations", 1000), "k_mccfr_resident_flat": ("trajectories", 1 << 20)}
res = {}
for kern, (unit, per_launch) in KERNELS.items():
    vals, dur = {}, []
    for f in glob.glob(os.path.join(out_dir, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(out_dir, "p1", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if not vals:
        continue
    mean = {k: sum(v[1:] or v) / len(v[1:] or v) for k, v in vals.items()}      # (the first launch of a pass warms up)
    res[kern] = {"unit": unit, "units_per_launch": per_launch, "counters_per_launch": mean,
                 "per_unit": {k: v / per_launch for k, v in mean.items()},
                 "launch_us_under_counters": [round(d, 1) for d in dur]}
path = os.path.join(ROOT, "profiles", f"{tag}_pmc_solvers.json")
res["source"] = (f"tools/pmc_solvers.sh {tag}: separate rocprofv3 --pmc passes (beside --kernel-trace only) over "
                 "tools/probe_solvers_once.py; means per launch without each pass's first launch")
with open(path, "w") as f:
    json.dump(res, f, indent=1, sort_keys=True)
for kern in KERNELS:
    if kern in res:
        print(kern, json.dumps(res[kern]["per_unit"], sort_keys=True))
        print("   launch us:", res[kern]["launch_us_under_counters"])

sys.path.insert(0, os.path.join(ROOT, "tools"))
import profile_sources  # the profile is stamped with the hashes of the sources that define its kernels
profile_sources.stamp(path, "pmc_solvers")
