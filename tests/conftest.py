import importlib
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pkg():
    return importlib.import_module("3dunetcnn_amd")


@pytest.fixture(scope="session")
def emu_backend():
    """Backend over the CPU-emulated build of the SAME kernel sources (tools/emu) -- test infrastructure only."""
    so = os.path.join(ROOT, "tools", "emu", "libmi355unet3d_emu.so")
    subprocess.check_call([os.path.join(ROOT, "tools", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import ctypes
    lib_mod = importlib.import_module("3dunetcnn_amd._lib")
    ops = importlib.import_module("3dunetcnn_amd.ops")
    lib = lib_mod.bind(ctypes.CDLL(so))
    return ops.Backend(lib=lib, device="cpu")


@pytest.fixture(scope="session")
def hip_backend():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ops = importlib.import_module("3dunetcnn_amd.ops")
    return ops.default_backend()
