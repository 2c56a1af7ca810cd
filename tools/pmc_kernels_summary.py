"""Per-kernel means of the counters collected by tools/pmc_kernels.sh (rocprofv3 *counter_collection.csv files)."""
import csv, glob, os, sys
out, key = sys.argv[1], sys.argv[2]
vals, dur = {}, {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if key not in name:
            continue
        short = name.split("(")[0][-70:]
        vals.setdefault((short, r["Grid_Size"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(out, "p1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if key in name:
            dur.setdefault((name.split("(")[0][-70:], r.get("Grid_Size", r.get("Grid_Size_X", "?"))), []).append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
kernels = sorted({(k[0], k[1]) for k in vals})
for kern in kernels:
    print(f"== {kern[0]}  grid {kern[1]}")
    for (short, grid), d in dur.items():
        if short == kern[0]:
            print(f"   duration under the counters (us): {[round(x, 1) for x in d[:6]]}")
    for (short, grid, counter), v in sorted(vals.items()):
        if (short, grid) == kern:
            v = v[len(v) // 3:]
            print(f"   {counter:42s} {sum(v) / len(v):18.1f}   ({len(v)} launches)")
