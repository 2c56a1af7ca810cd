#!/usr/bin/env python3
"""Are the committed counter profiles bench.py quotes still profiles of the CURRENT kernels?

bench.py's secondary.{mcts,cfr,mccfr}.roofline take instruction mixes from committed profile files (labelled `source`):
they go stale silently when a kernel changes.  Every summariser that writes such a profile (tools/pmc_solvers.py,
tools/pmc_mcts.py) stamps it with the sha256 of the sources that define the profiled kernel — the .hip file and every
header of open_spiel_amd/csrc — in a sidecar <profile>.sources.json; this tool compares the stamp of the NEWEST profile of
each kind with the files as they are now (content, not time stamps: works in a snapshot without .git).

  python tools/profile_sources.py            # table; exit 1 if a quoted profile is stale or unstamped
  python tools/profile_sources.py --stamp profiles/r06_pmc_solvers.json pmc_solvers     # what the summarisers call

tools/gpu_validation.sh runs the check after it has regenerated the profiles; tests/test_bench_line.py runs it on the CPU.
"""
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "open_spiel_amd", "csrc")
# kind -> (glob of the profiles bench.py reads, the .hip files that define the profiled kernels)
KINDS = {
    "pmc_solvers": ("r*_pmc_solvers.json", ["osg_cfr_small.hip", "osg_cfr_mccfr.hip"]),   # k_cfr_small, k_mccfr_resident_flat
    "pmc_k_mcts_wave": ("r*_pmc_k_mcts_wave_hex9_8192x1024.csv", ["osg_mcts_wave.hip"]),   # k_mcts_wave<HexT<3>, ...>
}


def source_hashes(hip_files):
    files = [os.path.join(CSRC, f) for f in hip_files] + sorted(glob.glob(os.path.join(CSRC, "*.h")))
    out = {}
    for f in files:
        with open(f, "rb") as fh:
            out[os.path.relpath(f, ROOT)] = hashlib.sha256(fh.read()).hexdigest()
    return out


def stamp(profile_path, kind):
    rec = {"kind": kind, "profile": os.path.relpath(os.path.abspath(profile_path), ROOT), "sources": source_hashes(KINDS[kind][1])}
    with open(profile_path + ".sources.json", "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    return rec


def newest(kind):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", KINDS[kind][0])))
    return files[-1] if files else None


def status(kind):
    """(profile or None, "current" | "stale: <files>" | "unstamped" | "missing")."""
    prof = newest(kind)
    if prof is None:
        return None, "missing"
    side = prof + ".sources.json"
    if not os.path.exists(side):
        return prof, "unstamped"
    with open(side) as f:
        then = json.load(f)["sources"]
    now = source_hashes(KINDS[kind][1])
    changed = sorted(k for k in now if then.get(k) != now[k])
    return prof, ("current" if not changed else "stale: " + ", ".join(changed))


def is_current(profile_relpath):
    """For bench.py: True / False for a profile it quotes (None if the profile is of no stamped kind)."""
    for kind in KINDS:
        prof, st = status(kind)
        if prof and os.path.relpath(prof, ROOT) == profile_relpath:
            return st == "current"
    return None


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--stamp":
        print(json.dumps(stamp(sys.argv[2], sys.argv[3]))[:200])
        return 0
    bad = 0
    for kind in KINDS:
        prof, st = status(kind)
        print(f"{kind:18s} {os.path.relpath(prof, ROOT) if prof else '-':60s} {st}")
        bad += st != "current"
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
