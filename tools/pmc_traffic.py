"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of
`bench.py --no-secondary --no-cpu-baseline` into profiles/pmc_traffic.json and the per-counter summaries.
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled for this kernel's 16-byte-per-lane reads as
MI355X_MICROARCH.md prescribes for gfx950.
  python tools/pmc_traffic.py gpurun_out/pmcF gpurun_out/pmcW [round-tag, default r01]
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_step_c4x2"
TAG = sys.argv[3] if len(sys.argv) > 3 else "r01"
out = {}
for d, counter in zip(sys.argv[1:3], ("FETCH_SIZE", "WRITE_SIZE")):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    vals = vals[len(vals) // 10:]  # drop the warm-up launches
    out[counter] = (len(vals), sum(vals) / len(vals), min(vals), max(vals))
    with open(os.path.join(ROOT, "profiles", f"{TAG}_pmc_{counter}_k_step_c4x2.csv"), "w") as f:
        f.write("Kernel_Name,Counter_Name,launches,mean_KB,min_KB,max_KB\n")
        f.write(f"\"k_step_c4x2<C4T<6,7,4>>\",{counter},{len(vals)},{out[counter][1]},{out[counter][2]},{out[counter][3]}\n")
fetch = out["FETCH_SIZE"][1] * 1024 * 2
write = out["WRITE_SIZE"][1] * 1024
res = {"kernel": "k_step_c4x2<C4T<6,7,4>>", "bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
       "raw": {"FETCH_SIZE_KB": out["FETCH_SIZE"][1], "WRITE_SIZE_KB": out["WRITE_SIZE"][1]},
       "algorithmic_bytes_per_launch": 35 << 20,
       "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, 2^20 states, {out['FETCH_SIZE'][0]} launches "
                 "each), profiles/{TAG}_pmc_*_k_step_c4x2.csv; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 16 B/lane reads)"}
json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
