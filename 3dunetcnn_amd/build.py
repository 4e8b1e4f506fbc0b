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
# conv3d_wino.hip: without SLP vectorisation. The fragment generation of conv3d_wino2d_d8 is written per element with a scalar sign; the SLP
# vectoriser packs part of it into v_pk_* (the sign then needs a register pair per lane) and hipcc unpacks some of that again in the shadow of
# the MFMAs -- a mix whose instruction count per phase the scheduler directives cannot name.
# conv3d_wino3d.hip: the same (its sched_group_barrier counts name scalar fp32 instructions).
EXTRA_FLAGS = {"conv3d_bf16_zring.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "conv3d_wino.hip": ["-fno-slp-vectorize"],
               "conv3d_wino3d.hip": ["-fno-slp-vectorize"]}
# what building WITHOUT an unsupported flag set costs, per file (printed with the warning)
EXTRA_FLAGS_LOST = {"conv3d_bf16_zring.hip": "the 16-bit plane-ring kernels then copy weight fragments between register halves: slower, same results",
                    "conv3d_wino.hip": "the scheduling directives of conv3d_wino2d_d8 assume the non-vectorised instruction mix: performance only",
                    "conv3d_wino3d.hip": "the scheduling directives of conv3d_wino3d assume the non-vectorised instruction mix: performance only"}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


_PROBED = {}


def flags_supported(cc, flags):
    """Does this hipcc accept `flags`? Compiles an empty translation unit once per flag set. -amdgpu-mfma-vgpr-form is an internal LLVM
    option: a ROCm whose LLVM does not know it aborts with 'Unknown command line argument', and the whole library -- including the fp32
    path, which does not need that file's kernels -- would fail to build."""
    key = tuple(flags)
    if key not in _PROBED:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "probe.hip")
            open(src, "w").write("// empty\n")
            r = subprocess.run([cc, "--offload-arch=gfx950", "-c", src, "-o", os.path.join(td, "probe.o")] + list(flags),
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            _PROBED[key] = r.returncode == 0
    return _PROBED[key]


def flags_for(cc, name, verbose=True):
    extra = EXTRA_FLAGS.get(name, [])
    if extra and not flags_supported(cc, extra):
        if verbose:
            print(f"build.py: WARNING: this hipcc does not accept {' '.join(extra)}; building {name} without it ({EXTRA_FLAGS_LOST.get(name, 'performance only')})",
                  file=sys.stderr, flush=True)
        extra = []
    return FLAGS + extra


def local_deps(depfile):
    """Headers of THIS tree an object was compiled from (its compiler-written depfile, -MD): editing one of them rebuilds the objects that
    include it and no others. No depfile yet: None (the caller falls back to every header)."""
    if not os.path.exists(depfile):
        return None
    root = os.path.abspath(os.path.join(HERE, ".."))
    words = open(depfile).read().replace("\\\n", " ").split()
    return [w for w in words[1:] if os.path.abspath(w).startswith(root) and os.path.exists(w)]


def build(force=False, verbose=True):
    srcs = sources()
    # without a depfile: every header a kernel source may include
    all_headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    all_headers.append(os.path.join(HERE, "..", "include", "mi355_unet3d.h"))
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        # the flag set is part of an object's identity: recorded beside it, a change rebuilds (mtimes alone would keep the old object)
        fl = flags_for(cc, os.path.basename(s), verbose)
        stamp = o + ".flags"
        same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(fl)
        deps = local_deps(o + ".d")
        if deps is None:
            deps = all_headers
        if force or not same_flags or newer(s, o) or any(newer(d, o) for d in deps):
            jobs.append(([cc] + fl + ["-MD", "-MF", o + ".d", "-c", s, "-o", o], stamp, " ".join(fl)))

    def run(job):
        cmd, stamp, text = job if isinstance(job, tuple) else (job, None, None)
        if verbose:
            print(" ".join(cmd), flush=True)
        if stamp and os.path.exists(stamp):
            os.remove(stamp)
        subprocess.check_call(cmd)
        if stamp:
            open(stamp, "w").write(text)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
