"""Summarise rocprofv3 --pmc counter_collection CSVs (one counter per pass) per kernel name.

    python tools/pmc_summary.py gpurun_out/<tag>/pmc_fetch/bench_counter_collection.csv gpurun_out/<tag>/pmc_write/bench_counter_collection.csv

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on
gfx950 counts 128-B requests as 64 B for wide coalesced reads: the `fetch_x2` column applies that correction. Both counters were
calibrated against 1 GiB moved in this library's access patterns (tools/micro/pmc_calibrate.hip, profiles/r4_pmc_calibration.txt):
FETCH_SIZE x 2 is exact for 16-byte and 4-byte-per-lane reads, WRITE_SIZE is exact.
"""
import csv
import hashlib
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """sha256 over the kernel sources (csrc/*.hip, *.h, in name order): bench.py refuses a traffic file collected from other kernels."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "3dunetcnn_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def tree_sha():
    """Commit the measured tree belongs to. The GPU box has no .git (gpurun ships the tree without it): tools/stamp_tree.sh writes
    `git rev-parse HEAD` (+ "-dirty" with uncommitted changes) into gpurun_stamp.txt before a call; with a .git present git answers."""
    if os.environ.get("MI355_GIT_SHA"):
        return os.environ["MI355_GIT_SHA"]
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
        if sha:
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True).stdout.strip()
            return sha + ("-dirty" if dirty else "")
    except OSError:
        pass
    stamp = os.path.join(ROOT, "gpurun_stamp.txt")
    if os.path.exists(stamp):
        return open(stamp).read().strip() or "unknown"
    return "unknown"


def load(path):
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    name = None
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Counter_Name"]
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return name, agg


def main(paths):
    tables = dict(load(p) for p in paths)
    names = set()
    for t in tables.values():
        names |= set(t)
    # GRBM_GUI_ACTIVE (optional third pass): busy cycles of the kernel summed over the 8 XCDs -> the shader clock the kernel ran at
    # (MI355X_MICROARCH.md, DVFS: the chip clocks to its power budget; a profiled pass runs a few per cent below an un-profiled one)
    clk = tables.pop("GRBM_GUI_ACTIVE", None)
    cols = sorted(tables)
    print(f"# provenance: git_sha={tree_sha()} kernel_source_sha256={kernel_source_hash()}")
    print("Kernel,Calls," + ",".join(f"{c}_MiB_per_launch" for c in cols) + ",fetch_x2_MiB_per_launch,avg_ms_under_pmc" + (",shader_clock_ghz" if clk else ""))
    rows = []
    for n in names:
        calls = max(tables[c][n][0] for c in cols if n in tables[c])
        vals = [tables[c][n][1] / max(tables[c][n][0], 1) / 1024.0 if n in tables[c] else float("nan") for c in cols]
        secs = max(tables[c][n][2] / max(tables[c][n][0], 1) for c in cols if n in tables[c])
        fx2 = 2 * vals[cols.index("FETCH_SIZE")] if "FETCH_SIZE" in cols else float("nan")
        tot = sum(tables[c][n][2] for c in cols if n in tables[c])
        ghz = clk[n][1] / 8.0 / clk[n][2] / 1e9 if clk and n in clk and clk[n][2] > 0 else float("nan")
        rows.append((tot, n, calls, vals, fx2, secs, ghz))
    for tot, n, calls, vals, fx2, secs, ghz in sorted(rows, reverse=True):
        print(f"\"{n}\",{calls}," + ",".join(f"{v:.2f}" for v in vals) + f",{fx2:.2f},{secs * 1e3:.4f}" + (f",{ghz:.3f}" if clk else ""))


if __name__ == "__main__":
    main(sys.argv[1:])
