"""Summarises the three rocprofv3 --pmc passes of tools/probe_mcts_one.py (hex(9), 8192 roots x 1024 simulations,
wave-per-root kernel) into profiles/<tag>_pmc_k_mcts_wave_hex9_8192x1024.csv: counter totals and per-simulation
values.   python tools/pmc_mcts.py gpurun_out/<tag> <tag>"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir, tag = sys.argv[1], sys.argv[2]
sims = 8192 * 1024
tot = {}
# one file per counter pass: the newest, should a directory hold the files of an earlier pass with the same tag
files = []
for d in sorted(glob.glob(os.path.join(out_dir, "pmc_mcts_*"))):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if fs:
        files.append(max(fs, key=os.path.getmtime))
for f in files:
    for r in csv.DictReader(open(f)):
        if "k_mcts_wave" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
path = os.path.join(ROOT, "profiles", f"{tag}_pmc_k_mcts_wave_hex9_8192x1024.csv")
with open(path, "w") as f:
    f.write('Counter,Value,Per_simulation,"note: k_mcts_wave<HexT<3>,true,true,false>, one search of 8192 roots x 1024 sims '
            '(8.39e6 simulations), separate rocprofv3 --pmc passes"\n')
    for k in sorted(tot):
        f.write(f"{k},{tot[k]:.0f},{tot[k] / sims:.1f}\n")
print(open(path).read())

sys.path.insert(0, os.path.join(ROOT, "tools"))
import profile_sources  # the profile is stamped with the hashes of the sources that define its kernels
profile_sources.stamp(path, "pmc_k_mcts_wave")
