"""Prints VGPR / AGPR / occupancy / LDS / spills per kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py 3dunetcnn_amd/csrc/conv3d_fwd.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark: +(VGPRs|AGPRs|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" ")[0] + ("Spill" if "Spill" in m.group(1) else "")] = int(m.group(2))
for r in rows:
    if flt in r["name"]:
        print(f'{r["name"][:110]:110s} V{r.get("VGPRs", 0):4d} A{r.get("AGPRs", 0):4d} occ{r.get("Occupancy", 0):2d} lds{r.get("LDS", 0):6d} '
              f'spill{r.get("VGPRsSpill", 0)} scratch{r.get("ScratchSize", 0)}')
