import importlib
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"` on a box without a GPU) runs the kernels on the emulator: ~15 minutes on 8 workers, about an hour on
    one. A whole-directory run without a GPU and without an explicit -n therefore distributes itself over pytest-xdist workers (when the
    plugin is there); single files / node ids, `-n ...`, a visible GPU, or MI355_TEST_SERIAL=1 leave the run as invoked."""
    opt = config.option
    if (hasattr(config, "workerinput") or os.environ.get("MI355_TEST_SERIAL") or not config.pluginmanager.hasplugin("xdist")
            or getattr(opt, "numprocesses", None) is not None or getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False)
            or getattr(opt, "keyword", "") or torch.cuda.is_available()):
        return None
    args = [a for a in config.args if "::" not in a]
    if len(args) != len(config.args) or not all(os.path.isdir(a) for a in args):
        return None
    opt.numprocesses = min(8, os.cpu_count() or 1)
    return None


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
