"""Winograd F(2x2, 3x3) x direct-z form of the 3x3x3 stride-1 convolution (csrc/conv3d_wino.hip) on the CPU emulator, against
F.conv3d: fp32, tolerance 1e-5 (measured ~1e-6). The emulator checks the index logic and the order of operations; the GPU twins
(tests/test_wino_gpu.py) run the same cases through the HIP library."""
import pytest
import torch
import torch.nn.functional as F

import op_cases as C
from oracle import torch_ops as O

ops = C.ops


@pytest.fixture(params=["2d", "3d"])
def wino_form(emu_backend, request):
    """Both Winograd forward / dgrad kernels take every case of this file: "2d" = conv3d_wino2d_d8 (F(2x2, 3x3) x direct z), "3d" =
    conv3d_wino3d (F(2x2x2, 3x3x3), round 6). The backend's form decides which pack wino_pack_weight builds and which kernel reads it."""
    old = emu_backend.wino_form
    emu_backend.wino_form = request.param
    yield request.param
    emu_backend.wino_form = old


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=8, cout=32, dhw=(2, 8, 16)),                                   # exactly one tile
    dict(n=2, cin=32, cout=32, dhw=(4, 8, 16), bias=True),                       # two z tiles, four channel chunks
    dict(n=1, cin=16, cout=64, dhw=(3, 9, 19), norm=True, residual=True, chscale=True),      # ragged in every axis, two channel tiles
    dict(n=1, cin=12, cout=40, dhw=(5, 6, 7), norm=True, slope=0.01, bias=True),  # partial channel chunk and tile
    dict(n=2, cin=8, cout=32, dhw=(8, 8, 16), bias=True),                        # 8 workgroups: the XCD-contiguous workgroup order
    dict(n=1, cin=16, cout=64, dhw=(8, 8, 16), norm=True),                       # the same with two channel tiles (4 spatial groups)
    dict(n=1, cin=8, cout=32, dhw=(16, 16, 32), bias=True),                      # 8 z tiles: the z-brick workgroup order (32 workgroups)
    # round 5 (conv3d_wino2d_d8: every staged slot is a DMA request): a single plane (three of the four requested planes lie outside the
    # volume: zeros from the device constant), an image smaller than the tile (whole request waves fetch zeros), three samples with a
    # partial channel chunk, and a plain input with every epilogue operand
    dict(n=1, cin=8, cout=8, dhw=(1, 3, 5), bias=True),
    dict(n=1, cin=16, cout=32, dhw=(2, 2, 3), norm=True),
    dict(n=3, cin=20, cout=32, dhw=(3, 4, 33), norm=True, residual=True, bias=True),
    dict(n=2, cin=40, cout=72, dhw=(5, 8, 16), residual=True, chscale=True, bias=True),
])
def test_wino_forward_matches_conv3d(emu_backend, kw, wino_form):
    be = emu_backend
    n, cin, cout, dhw = kw["n"], kw["cin"], kw["cout"], kw["dhw"]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, *dhw, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (cin * 27) ** 0.5)
    normspec = gamma = beta = None
    if kw.get("norm"):
        groups = 4
        gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
        normspec = (groups, gamma, beta, 1e-5, kw.get("slope", 0.0))
    b = torch.randn(cout, generator=g) if kw.get("bias") else None
    res = torch.randn(n, cout, *dhw, generator=g) if kw.get("residual") else None
    cs = (torch.rand(n, cout, generator=g) > 0.3).float() * 1.25 if kw.get("chscale") else None
    ref = O.conv_block(x, wt, 1, 1, normspec, b, res, cs)
    xa, ya = C.to_act(be, x), C.to_act(be, torch.zeros_like(ref))
    up = be.wino_pack_weight(wt, 0)
    extra = {}
    if normspec:
        mr, sc, sh = be.gn_stats(xa, normspec[0], 1e-5, gamma, beta)
        extra = dict(in_mode=ops.IN_AFFINE_ACT, slope=normspec[4], scale=sc, shift=sh)
    be.conv_fwd_wino(xa, up, ya, bias=b, residual=C.to_act(be, res) if res is not None else None, chscale=cs, **extra)
    assert C.rel_err(C.from_act(ya), ref) < 1e-5


def test_wino_dgrad_pack_matches_autograd(emu_backend, wino_form):
    be = emu_backend
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 4, 8, 16, generator=g, requires_grad=True)
    wt = torch.randn(32, 16, 3, 3, 3, generator=g) * 0.05
    y = F.conv3d(x, wt, padding=1)
    dy = torch.randn(y.shape, generator=g)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    dya, dxa = C.to_act(be, dy), C.to_act(be, torch.zeros_like(x.detach()))
    be.conv_fwd_wino(dya, be.wino_pack_weight(wt, 1), dxa)
    assert C.rel_err(C.from_act(dxa), dx_ref) < 1e-5


class _WinoBackend:
    """The emulator backend with its 3x3x3 stride-1 convolutions redirected to the Winograd kernel, so that the shared op cases
    (tests/op_cases.py: epilogue fusions, concat slices, fused statistics) run against it unchanged."""
    def __init__(self, be):
        self._be = be
        self.wino_calls = 0

    def __getattr__(self, name):
        return getattr(self._be, name)

    def pack_weight(self, w, mode, *a, **k):
        if w.shape[2:] == (3, 3, 3) and mode in (0, 1):
            return ("wino", self._be.wino_pack_weight(w, mode), self._be.pack_weight(w, mode, *a, **k))
        return self._be.pack_weight(w, mode, *a, **k)

    def conv_fwd(self, x, wp, y, kd, stride=1, pad=None, **kw):
        if isinstance(wp, tuple):
            if kd == 3 and stride == 1 and kw.get("off", (0, 0, 0)) == (0, 0, 0) and kw.get("in_mode", ops.IN_PLAIN) in (ops.IN_PLAIN, ops.IN_AFFINE_ACT):
                kw.pop("off", None); kw.pop("out_dhw", None)
                self.wino_calls += 1
                return self._be.conv_fwd_wino(x, wp[1], y, **kw)
            wp = wp[2]
        return self._be.conv_fwd(x, wp, y, kd, stride, pad, **kw)


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=32, dhw=(3, 5, 19), norm=True, residual=True, chscale=True),
    dict(n=1, cin=8, cout=64, dhw=(4, 4, 16), norm=True, yld=128, yc0=32),                   # concat slice of a wider buffer
    dict(n=2, cin=4, cout=32, dhw=(3, 4, 17), bias=True),
    dict(n=1, cin=48, cout=40, dhw=(2, 6, 9), norm=True, slope=0.01),
])
def test_wino_shared_forward_cases(emu_backend, kw, wino_form):
    be = _WinoBackend(emu_backend)
    assert C.case_conv_fwd(be, **kw) < 1e-5
    assert be.wino_calls == 1


@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=64, dhw=(3, 4, 18)), dict(n=2, cin=64, cout=32, dhw=(4, 4, 16))])
def test_wino_shared_dgrad_cases(emu_backend, kw, wino_form):
    assert C.case_conv_dgrad(_WinoBackend(emu_backend), **kw) < 1e-5


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=32, dhw=(3, 5, 19), residual=True, chscale=True),
    dict(n=1, cin=16, cout=64, dhw=(4, 8, 16), yld=128, yc0=32),
    dict(n=2, cin=16, cout=40, dhw=(2, 6, 9), groups_out=40),
])
def test_wino_epilogue_moments(emu_backend, kw, wino_form):
    assert C.case_conv_moments(_WinoBackend(emu_backend), **kw) < 2e-5


@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=32, dhw=(3, 5, 19)), dict(n=2, cin=64, cout=16, dhw=(4, 4, 16), slope=0.01)])
def test_wino_norm_backward_sums_from_dgrad_epilogue(emu_backend, kw, wino_form):
    be = _WinoBackend(emu_backend)
    r = C.case_gn_bwd_fused(be, **kw)
    assert all(v < 2e-4 for v in r.values()), r
    assert be.wino_calls == 1


def test_whole_network_step_on_the_winograd_kernels(emu_backend, wino_form):
    """Product routing with the size threshold removed: every eligible 3x3x3 stride-1 forward / dgrad / wgrad conv of a UNet3D step on the
    Winograd kernels, against the golden bundle generated from the reference."""
    import importlib
    import os
    unet = importlib.import_module("3dunetcnn_amd.unet")
    losses = importlib.import_module("3dunetcnn_amd.losses")
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "unet3d_small.pt"))
    be = emu_backend
    old_routing = (be.winograd, be.wgrad_form)
    be.winograd, be.wgrad_form = True, "wino"
    be.WINO_MIN_VOXELS = 0                 # the golden bundle is 20 x 16 x 24: route every level
    calls = {"n": 0}
    orig = be.conv_fwd_wino

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    be.conv_fwd_wino = counted
    try:
        m = unet.HipUNet3D(**g["kwargs"]).eval()
        m._be = be
        m.load_state_dict(g["state_dict"])
        crit = losses.HipDiceLoss(sigmoid=True)
        crit._be = be
        out = m(g["x"])
        loss = crit(out, g["y"])
        loss.backward()
    finally:
        be.winograd, be.wgrad_form = old_routing
        del be.conv_fwd_wino, be.WINO_MIN_VOXELS
    assert calls["n"] >= 20
    assert C.rel_err(out, g["logits"]) < 1e-3
    assert abs(float(loss.detach()) - float(g["loss"])) / float(g["loss"]) < 1e-3
    for k, p in m.named_parameters():
        assert C.rel_err(p.grad, g["grads"][k]) < 1e-3, k


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=32, dhw=(1, 4, 8)),                                   # one plane, exactly one plane tile
    dict(n=2, cin=32, cout=32, dhw=(3, 8, 8), norm=True),                        # two tile rows, the ring wraps once
    dict(n=1, cin=8, cout=64, dhw=(4, 9, 17)),                                   # ragged plane tiles, two co tiles, partial ci tile
    dict(n=1, cin=40, cout=96, dhw=(5, 3, 7), norm=True, slope=0.01),            # partial channel tiles both sides
    dict(n=2, cin=32, cout=32, dhw=(7, 16, 32), norm=True),                      # several columns per workgroup, ring wraps twice
])
def test_wgrad_wino_ring_matches_autograd(emu_backend, kw):
    """The plane-ring Winograd weight gradient (csrc/conv3d_wgrad_wino.hip) on the shared weight-gradient case."""
    be = emu_backend
    old = (be.wgrad_form, getattr(be, "WINO_MIN_VOXELS"))
    be.wgrad_form, be.WINO_MIN_VOXELS = "wino", 0
    calls = {"n": 0}
    orig = be.lib.mi355_conv3d_wgrad_wino

    def counted(*a):
        calls["n"] += 1
        return orig(*a)
    be.lib.mi355_conv3d_wgrad_wino = counted
    try:
        assert C.case_conv_wgrad(be, **kw) < 1e-4
    finally:
        be.lib.mi355_conv3d_wgrad_wino = orig
        be.wgrad_form = old[0]
        del be.WINO_MIN_VOXELS
    assert calls["n"] == 1


def test_ineligible_call_falls_back_to_the_direct_kernel(emu_backend):
    """Routing policy (size, channels) lives in ops.Backend.conv_fwd; ELIGIBILITY is the library's answer (mi355_conv3d_wino_supported).
    A call the Winograd kernel cannot take -- here an input view whose leading dimension is not a multiple of 4 channels is not
    constructible, so: an input Act at a 8-byte-aligned (not 16-byte) channel offset is refused by Act itself; the reachable case is a
    residual narrower than the output -- must run (on the direct kernel) and give the direct kernel's result, not raise."""
    import ctypes
    be = emu_backend
    assert be.winograd
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 16, 16, 16, generator=g)
    w = torch.randn(16, 16, 3, 3, 3, generator=g) * 0.1
    xa, ya = C.to_act(be, x), C.to_act(be, torch.zeros(1, 16, 16, 16, 16))
    d = be._desc(3, 1, 1, 0, 0.0, None, None, None, None, None, (0, 0, 0), (16, 16, 16), [])
    xd, yd = xa.desc(), ya.desc()
    assert be.lib.mi355_conv3d_wino_supported(ctypes.byref(xd), ctypes.byref(yd), ctypes.byref(d)) == 1
    d.off_x = 1                                              # a windowed output: not a Winograd call
    assert be.lib.mi355_conv3d_wino_supported(ctypes.byref(xd), ctypes.byref(yd), ctypes.byref(d)) == 0
    d.off_x = 0
    d.stride = 2
    assert be.lib.mi355_conv3d_wino_supported(ctypes.byref(xd), ctypes.byref(yd), ctypes.byref(d)) == 0
    # end to end: the library refuses -> conv_fwd runs the direct kernel and returns its result
    calls = []
    real = be.lib.mi355_conv3d_wino_supported
    try:
        be.lib.mi355_conv3d_wino_supported = lambda *a: (calls.append(1), 0)[1]
        be.conv_fwd(xa, be.pack_weight(w, 0), ya, 3, 1)
    finally:
        be.lib.mi355_conv3d_wino_supported = real
    assert calls and C.rel_err(C.from_act(ya), torch.nn.functional.conv3d(x, w, padding=1)) < 1e-5
