"""Builds libmi355unet3d.so in-tree with hipcc for gfx950 (the only supported target).

    python 3dunetcnn_amd/build.py [--force]

hipcc cross-compiles without a GPU. The .so is git-ignored but travels with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmi355unet3d.so")
OBJ = os.path.join(HERE, "csrc", "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Per-source flags. The plane-ring kernels pin 216 weight registers in the AGPRs; left to its heuristics hipcc puts the MFMA accumulators
# there as well, runs out of AGPRs, keeps the remaining weights in VGPRs and copies each into place before the MFMA that reads it (two
# v_mov_b64 + s_nop per MFMA on a third of the MFMAs of the 64-channel form). -amdgpu-mfma-vgpr-form selects the VGPR-destination MFMA
# forms: accumulators in the architectural half, every weight fragment in an AGPR, no copies, no spills (tools/lpz_one.sh).
EXTRA_FLAGS = {"conv3d_bf16_zring.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    srcs = sources()
    # every header a kernel source may include: editing any of them rebuilds all objects
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    deps.append(os.path.join(HERE, "..", "include", "mi355_unet3d.h"))
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or newer(s, o) or any(newer(d, o) for d in deps):
            jobs.append([cc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
