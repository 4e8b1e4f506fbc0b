"""CPU suite: pins the oracle, the drop-in boundary and the host logic (no GPU needed).

* the oracle restatement (oracle/unet3d_ref.py) reproduces the committed golden vectors, which were produced by importing
  the REFERENCE UNet3D (oracle/make_golden.py); where /root/reference exists (build container) it is also compared live;
* HipUNet3D exposes the reference's state_dict keys / shapes and the same default initialisation under the same seed;
* the same kernel sources, run by the CPU emulator, reproduce the golden vectors end to end (fwd, loss, all grads);
* libmi355unet3d.so loads and exports every symbol declared in include/mi355_unet3d.h.
"""
import ctypes
import importlib
import os
import re

import pytest
import torch

import op_cases as C
from oracle import reference_shim, torch_ops as O, unet3d_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
unet = importlib.import_module("3dunetcnn_amd.unet")
losses = importlib.import_module("3dunetcnn_amd.losses")
lib_mod = importlib.import_module("3dunetcnn_amd._lib")


def _enc(kwargs):
    return tuple(kwargs.get("encoder_blocks") or (1, 2, 2, 4))


@pytest.mark.parametrize("name", ["unet3d_small.pt", "unet3d_small_transposed.pt"])
def test_oracle_reproduces_golden(name):
    g = torch.load(os.path.join(GOLD, name))
    kw = g["kwargs"]
    sd = {k: v.clone().requires_grad_(True) for k, v in g["state_dict"].items()}
    out = R.unet3d_forward(sd, g["x"], _enc(kw), None, kw.get("use_transposed_convolutions", False))
    loss = O.dice_loss(out, g["y"])
    loss.backward()
    assert C.rel_err(out, g["logits"]) < 1e-6
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    for k, v in g["grads"].items():
        assert C.rel_err(sd[k].grad, v) < 1e-5, k


@pytest.mark.skipif(not reference_shim.available(), reason="/root/reference only exists in the build container")
@pytest.mark.parametrize("kw,dhw", [
    (dict(n_features=4, n_outputs=3, base_width=8), (16, 16, 16)),
    (dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1, 2], use_transposed_convolutions=True), (12, 16, 8)),
    (dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 2, 1]), (15, 13, 10)),   # odd sizes: crop path
])
def test_oracle_matches_live_reference(kw, dhw):
    ref = reference_shim.build_reference_unet3d(seed=11, **kw).eval()
    x, y = R.synthetic_case(2, 4, dhw)
    want = ref(x)
    O.dice_loss(want, y).backward()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in ref.state_dict().items()}
    got = R.unet3d_forward(sd, x, _enc(kw), None, kw.get("use_transposed_convolutions", False))
    O.dice_loss(got, y).backward()
    assert torch.equal(got, want)
    for k, p in ref.named_parameters():
        assert C.rel_err(sd[k].grad, p.grad) < 1e-6, k


@pytest.mark.skipif(not reference_shim.available(), reason="/root/reference only exists in the build container")
@pytest.mark.parametrize("kw", [
    dict(n_features=4, n_outputs=3),
    dict(n_features=4, n_outputs=3, use_transposed_convolutions=True),
    dict(n_features=4, n_outputs=3, base_width=16, encoder_blocks=[1, 2, 2, 2, 4]),
])
def test_state_dict_contract_and_same_seed_init(kw):
    ref = reference_shim.build_reference_unet3d(seed=1234, **kw)
    torch.manual_seed(1234)
    mine = unet.HipUNet3D(**kw)
    a, b = ref.state_dict(), mine.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
        assert torch.equal(a[k], b[k]), f"{k}: same seed must give the reference's default init"
    mine.load_state_dict(a, strict=True)
    ref.load_state_dict(b, strict=True)


def test_state_dict_keys_match_golden():
    g = torch.load(os.path.join(GOLD, "unet3d_small.pt"))
    m = unet.HipUNet3D(**g["kwargs"])
    assert list(m.state_dict().keys()) == list(g["state_dict"].keys())
    m.load_state_dict(g["state_dict"], strict=True)
    n_params = sum(p.numel() for p in unet.HipUNet3D(n_features=4, n_outputs=3).parameters())
    assert n_params == 23970216


@pytest.mark.parametrize("name", ["unet3d_small.pt", "unet3d_small_transposed.pt"])
def test_emulated_kernels_reproduce_golden(emu_backend, name):
    g = torch.load(os.path.join(GOLD, name))
    m = unet.HipUNet3D(**g["kwargs"]).eval()
    m._be = emu_backend                       # test-only injection of the CPU-emulated kernel library
    m.load_state_dict(g["state_dict"])
    out = m(g["x"])
    crit = losses.HipDiceLoss(sigmoid=True)
    crit._be = emu_backend
    loss = crit(out, g["y"])
    loss.backward()
    assert C.rel_err(out, g["logits"]) < 1e-3
    assert abs(float(loss.detach()) - float(g["loss"])) / float(g["loss"]) < 1e-3
    for k, p in m.named_parameters():
        assert C.rel_err(p.grad, g["grads"][k]) < 1e-3, k


def _autocast_golden(be, dev):
    """HipAutocastUNet against the REFERENCE graph run under torch.autocast (tests/golden/unet3d_small_autocast.pt, generated by
    oracle/make_golden.py from the imported reference; AutocastUNet = UNet3D.forward under autocast, unet.py:53-58). Two 16-bit
    roundings of the same fp32 network are compared, so the bound is the sum of both errors. The reference rounds every conv OUTPUT to
    16 bits; so do activation_storage="bf16" (the bf16 mode's default) and "fp16" (the reference's own amp form; values rounded once, on
    store), while activation_storage="fp32" keeps fp32 tensors between the convolutions: either way the HIP result must be at least as
    close to the fp32 logits as the reference's own autocast result is."""
    a = torch.load(os.path.join(GOLD, "unet3d_small_autocast.pt"))
    g = torch.load(os.path.join(GOLD, a["bundle"]))
    res = {}
    for mode, storage, tol in (("fp16", "fp32", 4e-3), ("fp16", "fp16", 4e-3), ("bf16", "bf16", 3e-2), ("bf16", "fp32", 3e-2)):
        m = unet.HipAutocastUNet(autocast_dtype=mode, activation_storage=storage, **g["kwargs"]).to(dev).eval()
        assert m.act_storage == {"bf16": torch.bfloat16, "fp16": torch.float16}.get(storage) and \
            unet.HipAutocastUNet(autocast_dtype=mode, **g["kwargs"]).act_storage == (torch.bfloat16 if mode == "bf16" else None)
        if be is not None:
            m._be = be
        m.load_state_dict(g["state_dict"])
        with torch.no_grad():
            out = m(g["x"].to(dev)).cpu()
        ref16, ref32 = a["logits_" + mode], g["logits"]
        e_ref = C.rel_err(ref16, ref32)                      # the reference's own autocast error
        e_hip = C.rel_err(out, ref32)
        assert C.rel_err(out, ref16) < tol, (mode, C.rel_err(out, ref16))
        assert 1e-6 < e_hip <= e_ref, (mode, storage, e_hip, e_ref)  # really the 16-bit path, and no worse than the reference's
        res[mode] = max(res.get(mode, 0.0), e_hip)
        res[mode + "/" + storage] = e_hip
    assert res["fp16"] < 0.25 * res["bf16"]                  # 11 vs 8 significand bits
    return res


def test_autocast_unet_matches_reference_under_autocast_on_emulator(emu_backend):
    _autocast_golden(emu_backend, "cpu")


@pytest.mark.gpu
def test_autocast_unet_matches_reference_under_autocast_gpu():
    print(_autocast_golden(None, "cuda"))


def test_mixed_precision_gradients_against_the_reference_gradients(emu_backend):
    """Whole-network gradients of the 16-bit modes against the REFERENCE's fp32 gradients of the golden bundle (dgrad and wgrad run on
    the 16-bit pipe too). A freshly initialised norm + Dice network amplifies operand rounding ~40x, so the bound is
    global: relative error of the concatenated gradient and its cosine to the reference's -- measured fp16 2.0e-2 / 1 - 2e-4,
    bf16 6.1e-2 / 1 - 1.8e-3."""
    g = torch.load(os.path.join(GOLD, "unet3d_small.pt"))
    res = {}
    for mode, storage, tol, cos_min in (("fp16", "fp32", 5e-2, 0.999), ("bf16", "fp32", 1.5e-1, 0.99), ("bf16", "bf16", 1.5e-1, 0.99)):
        m = unet.HipAutocastUNet(autocast_dtype=mode, activation_storage=storage, **g["kwargs"]).eval()
        m._be = emu_backend
        m.load_state_dict(g["state_dict"])
        crit = losses.HipDiceLoss(sigmoid=True)
        crit._be = emu_backend
        loss = crit(m(g["x"]), g["y"])
        loss.backward()
        a = torch.cat([p.grad.reshape(-1) for _, p in m.named_parameters()])
        b = torch.cat([g["grads"][k].reshape(-1) for k, _ in m.named_parameters()])
        rel = float((a - b).norm() / b.norm())
        cos = float((a * b).sum() / a.norm() / b.norm())
        assert abs(float(loss.detach()) - float(g["loss"])) / float(g["loss"]) < 1e-4
        assert rel < tol and cos > cos_min, (mode, storage, rel, cos)
        res[mode + "/" + storage] = rel
    print(res)
    assert res["fp16/fp32"] < 0.5 * min(res["bf16/fp32"], res["bf16/bf16"])


def test_fp16_mode_rounds_operands_like_tensor_half(emu_backend):
    """MI355_PREC_F16 = conv of the fp16-ROUNDED operands (round to nearest even, as tensor.half()) with exact products and fp32
    accumulation: against F.conv3d of the rounded tensors the difference is accumulation order only."""
    import torch.nn.functional as F
    be = emu_backend
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 32, 4, 6, 18, generator=g) * 3.0
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
    ref = F.conv3d(x.half().float(), w.half().float(), padding=1)
    be.set_precision("fp16")
    try:
        xa, ya = C.to_act(be, x), C.to_act(be, torch.zeros_like(ref))
        be.conv_fwd(xa, be.pack_weight(w, 0), ya, 3, 1)
        out = C.from_act(ya)
    finally:
        be.set_precision("fp32")
    assert C.rel_err(out, ref) < 2e-6
    assert C.rel_err(out, F.conv3d(x, w, padding=1)) > 1e-5        # and it is not the fp32 path


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mi355_unet3d.h")).read()
    declared = set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib_mod.SIGNATURES), declared ^ set(lib_mod.SIGNATURES)
    if not os.path.exists(lib_mod.LIB_PATH):
        importlib.import_module("3dunetcnn_amd.build").build(verbose=False)
    lib = ctypes.CDLL(lib_mod.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib_mod.bind(lib).mi355_version()


def test_product_path_refuses_cpu():
    m = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1])
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.randn(1, 4, 8, 8, 8))
    with pytest.raises(RuntimeError, match="MI355X"):
        losses.HipDiceLoss(sigmoid=True)(torch.randn(1, 3, 4, 4, 4), torch.zeros(1, 3, 4, 4, 4))


def test_optimizer_state_checkpoint_resumes_bitwise(emu_backend):
    """the reference checkpoints only model.state_dict() (train/train.py:86-89; Adam moments are lost on resume).
    Superset here: HipAdam.state_dict() round-trips, so a resumed run continues bit-identically."""
    import io
    optim = importlib.import_module("3dunetcnn_amd.optim")
    kw = dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1])
    x, y = R.synthetic_case(1, 4, (8, 8, 8), 3)

    def make():
        torch.manual_seed(0)
        m = unet.HipUNet3D(**kw).eval()
        m._be = emu_backend
        o = optim.HipAdam(m.parameters(), lr=1e-3)
        o._be = emu_backend
        c = losses.HipDiceLoss(sigmoid=True)
        c._be = emu_backend
        return m, o, c

    def step(m, o, c):
        o.zero_grad()
        l = c(m(x), y)
        l.backward()
        o.step()
        return float(l.detach())
    m, o, c = make()
    for _ in range(2):
        step(m, o, c)
    buf = io.BytesIO()
    torch.save({"model": m.state_dict(), "opt": o.state_dict()}, buf)
    step(m, o, c)
    m2, o2, c2 = make()
    ck = torch.load(io.BytesIO(buf.getvalue()))
    m2.load_state_dict(ck["model"])
    o2.load_state_dict(ck["opt"])
    step(m2, o2, c2)
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
