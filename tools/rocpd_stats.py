"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the classic --stats table.

    python tools/rocpd_stats.py gpurun_out/prof_r1/bench_results.db > profiles/r1_bench_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                      "max(accum_vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,VGPR,AGPR,LDS")
    for n, c, s, a, mn, mx, vg, ag, lds in rows:
        print(f"\"{n}\",{c},{s},{a:.0f},{100.0 * s / tot:.2f},{mn},{mx},{vg},{ag},{lds}")


if __name__ == "__main__":
    main(sys.argv[1])
