"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of
`bench.py --no-secondary --no-cpu-baseline` into profiles/pmc_traffic.json and the per-counter summaries.
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (the
128-byte requests of a coalesced streaming read are tallied at 64 B).  The guide calibrated that on 16-byte-per-lane
reads; this kernel reads 8 bytes per lane, so the same pass's plain copy kernel (k_copy16, whose bytes are known)
and the kernel's own algorithmic read bytes were compared with the raw counter: both show the factor 1/2.  The
bench launches the headline kernel at two sizes (2^20 states = Grid_Size 2^20 threads, and the 2^24-state DRAM
leg); they are reported separately
(the top-level keys describe the 2^20 launch, `dram_leg` the 2^24 one).
  python tools/pmc_traffic.py gpurun_out/pmcF gpurun_out/pmcW [round-tag, default r01]
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_step_c4std"
TAG = sys.argv[3] if len(sys.argv) > 3 else "r01"
by_grid = {}
for d, counter in zip(sys.argv[1:3], ("FETCH_SIZE", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
                # states per launch: k_step_c4std2 takes two per thread, k_step_c4std one
                states = int(r["Grid_Size"]) * (2 if "k_step_c4std2" in r["Kernel_Name"] else 1)
                by_grid.setdefault(states, {}).setdefault(counter, []).append(float(r["Counter_Value"]))
res = None
rows = []
for states in sorted(by_grid):
    grid = states
    stat = {}
    for counter, vals in by_grid[grid].items():
        vals = vals[len(vals) // 10:]  # drop the warm-up launches
        stat[counter] = (len(vals), sum(vals) / len(vals), min(vals), max(vals))
        rows.append((states, counter) + stat[counter])
    if "FETCH_SIZE" not in stat or "WRITE_SIZE" not in stat:
        continue
    fetch = stat["FETCH_SIZE"][1] * 1024 * 2
    write = stat["WRITE_SIZE"][1] * 1024
    rec = {"kernel": "k_step_c4std2 (k_step_c4std for odd batches)", "states": states, "bytes_per_launch": fetch + write,
           "fetch_bytes": fetch, "write_bytes": write,
           "raw": {"FETCH_SIZE_KB": stat["FETCH_SIZE"][1], "WRITE_SIZE_KB": stat["WRITE_SIZE"][1]},
           "algorithmic_bytes_per_launch": 35 * states,
           "ratio_to_algorithmic": (fetch + write) / (35 * states),
           "launches": stat["FETCH_SIZE"][0]}
    if states == 1 << 20 or res is None:
        dram = res.get("dram_leg") if res else None
        res = dict(rec)
        if dram:
            res["dram_leg"] = dram
    if states == 1 << 24:
        res["dram_leg"] = rec
res["source"] = (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of bench.py --no-secondary "
                 f"--no-cpu-baseline), profiles/{TAG}_pmc_*_k_step_c4std.csv; FETCH_SIZE doubled per "
                 "MI355X_MICROARCH.md (gfx950 tallies the 128-byte requests of coalesced reads at 64 B; checked in "
                 "the same pass against the plain copy k_copy16, whose bytes are known)")
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    with open(os.path.join(ROOT, "profiles", f"{TAG}_pmc_{counter}_k_step_c4std.csv"), "w") as f:
        f.write("Kernel_Name,states,Counter_Name,launches,mean_KB,min_KB,max_KB\n")
        for states, c, cnt, mean, lo, hi in rows:
            if c == counter:
                f.write(f"\"k_step_c4std\",{states},{counter},{cnt},{mean},{lo},{hi}\n")
json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
