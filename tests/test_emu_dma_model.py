"""Self-test of the CPU emulator's LDS-DMA model (tools/emu/emu.h): the emulator is what checks the counted-wait protocol of
conv3d_wino2d_d8 on the CPU (requests land only when a counted wait of their wave retires them; tests/test_wino_emu.py runs under this model
by default and once under MI355_EMU_DMA=early), so the model itself is held to its definition here."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def selftest(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emu") / "dma_model")
    emu = os.path.join(ROOT, "tools", "emu")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-I", emu, os.path.join(ROOT, "tests", "emu_selftest", "dma_model.cpp"),
                           os.path.join(emu, "emu.cpp"), "-lpthread", "-o", exe])
    return exe


@pytest.mark.parametrize("model", ["late", "early"])
def test_dma_model(selftest, model):
    env = dict(os.environ, MI355_EMU_DMA=model)
    p = subprocess.run([selftest], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert f"{model} model: 0 failures" in p.stdout


def test_wino_forward_under_the_early_model(monkeypatch):
    """The Winograd forward cases once more with requests that land at once (the WAR half: a buffer restaged while it is still read). A
    subprocess: the emulator reads MI355_EMU_DMA once."""
    env = dict(os.environ, MI355_EMU_DMA="early")
    p = subprocess.run(["python", "-m", "pytest", os.path.join(ROOT, "tests", "test_wino_emu.py"), "-q", "-x", "-k", "forward_matches or dgrad_pack"],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


def _variant_backend(tmp_path, tag, transform=None, defines=()):
    """An emulator library whose conv3d_wino.hip is a mutated copy (source transform and / or -D switches); the other objects are the
    emulator build's."""
    import ctypes
    import importlib
    csrc = os.path.join(ROOT, "3dunetcnn_amd", "csrc")
    src = open(os.path.join(csrc, "conv3d_wino.hip")).read()
    if transform:
        src = transform(src)
    cpp = tmp_path / f"conv3d_wino_{tag}.cpp"
    cpp.write_text(src.replace('"../../include/mi355_unet3d.h"', f'"{ROOT}/include/mi355_unet3d.h"'))
    emu = os.path.join(ROOT, "tools", "emu")
    obj, so = str(tmp_path / f"wino_{tag}.o"), str(tmp_path / f"lib{tag}.so")
    subprocess.check_call(["g++", "-std=c++20", "-O2", "-fPIC", "-DMI355_EMU", "-w", "-I", emu, "-I", csrc, "-x", "c++", "-c", str(cpp), "-o", obj] + list(defines))
    others = [os.path.join(emu, "build", f) for f in os.listdir(os.path.join(emu, "build")) if f.endswith(".o") and f != "conv3d_wino.o"]
    subprocess.check_call(["g++", "-shared", "-o", so] + others + [obj, "-lpthread"])
    lib_mod = importlib.import_module("3dunetcnn_amd._lib")
    ops = importlib.import_module("3dunetcnn_amd.ops")
    return ops.Backend(lib=lib_mod.bind(ctypes.CDLL(so)), device="cpu")


def test_wrong_counted_waits_are_caught_on_the_cpu(emu_backend, tmp_path):
    """Mutation test: what makes the green runs of tests/test_wino_emu.py evidence for the wait immediates conv3d_wino2d_d8 carries. On the
    GPU a kernel that waits too little passes whenever the DMA happens to be fast enough; under the emulator's late model it must fail.
      never : the loop never waits for a request (the kernel's own timing ablation, -DWINO_ABL=32)
      lax   : every wait of the loop leaves five more instructions outstanding (about two phases of requests)
      slack : ONE wait (end of phase 1) one instruction too lax -- every request is waited for at least one phase before it is needed, so
              this one is still correct: the protocol has that much slack, and the test says so."""
    import re
    import torch
    import torch.nn.functional as F
    import op_cases as C

    def lax(src):
        out, n = re.subn(r"      D8_PHASE_END\(([13])\);", lambda m: f"      D8_PHASE_END({int(m.group(1)) + 5});", src)      # the four waits of the loop
        assert n == 4
        return out

    def slack(src):
        assert src.count("      D8_PHASE_END(1);\n") == 1
        return src.replace("      D8_PHASE_END(1);\n", "      D8_PHASE_END(2);\n")

    bes = {"product": emu_backend, "never": _variant_backend(tmp_path, "never", defines=["-DWINO_ABL=32"]),
           "lax": _variant_backend(tmp_path, "lax", lax), "slack": _variant_backend(tmp_path, "slack", slack)}
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 24, 4, 8, 16, generator=g)            # three channel chunks
    w = torch.randn(32, 24, 3, 3, 3, generator=g) * 0.05
    ref = F.conv3d(x, w, padding=1)
    errs = {}
    for name, be in bes.items():
        xa, ya = C.to_act(be, x), C.to_act(be, torch.zeros_like(ref))
        be.conv_fwd_wino(xa, be.wino_pack_weight(w, 0), ya)
        errs[name] = C.rel_err(C.from_act(ya), ref)
    assert errs["product"] < 1e-5 and errs["slack"] < 1e-5 and errs["never"] > 1e-2 and errs["lax"] > 1e-2, errs
