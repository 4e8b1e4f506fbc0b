"""Print per-kernel averages of every counter in a rocprofv3 counter_collection.csv (kernels matching a substring)."""
import csv, sys
from collections import defaultdict
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "conv3d")
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if pat not in k:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k in acc:
    print(k[:110], "avg_us", round(sum(dur[k]) / len(dur[k]), 1))
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} {sum(v) / len(v):.4g}")
