"""Full-size checks (BASELINE configs[1]/[4] sizes: 128^3 patches, batch 2; 240x240x155 volume) through size-independent
properties, because the CPU oracle needs minutes per step at these sizes:

  * adjoint identities  <conv(x; w), dy> == <x, dgrad(dy; w)> == <w, wgrad(x, dy)>   (all three kernels, one scalar each);
  * linearity in the weights;
  * two independent implementations agree: the exact-fp32 MFMA kernels vs the 6-product split-bf16 kernels (different
    instruction, tiling, LDS layout and lane maps) on the same 128^3 tensors;
  * batch independence and run-to-run bitwise determinism of the whole network step at 128^3, batch 2;
  * GroupNorm statistics of a tensor with known per-channel mean / variance; Dice of a perfect prediction;
  * sliding-window partition of unity on the 240x240x155 volume (an identity "network" must come back exactly).
"""
import importlib

import pytest
import torch

import op_cases as C
from oracle import unet3d_ref as R

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dunetcnn_amd.ops")
unet = importlib.import_module("3dunetcnn_amd.unet")
losses = importlib.import_module("3dunetcnn_amd.losses")
inferer = importlib.import_module("3dunetcnn_amd.inferer")
S = 128


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _conv_triplet(be, n, cin, cout, norm):
    g = torch.Generator(device="cuda").manual_seed(cin * 1000 + cout)
    x = be.empty_act(n, S, S, S, cin); x.buf.normal_(generator=g)
    dy = be.empty_act(n, S, S, S, cout); dy.buf.normal_(generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda", generator=g) * (1.0 / (27 * cin) ** 0.5)
    kw = {}
    if norm:
        gamma = torch.rand(cin, device="cuda", generator=g) + 0.5
        beta = torch.randn(cin, device="cuda", generator=g) * 0.3
        groups = 8 if cin % 8 == 0 else cin
        mr, sc, sh = be.gn_stats(x, groups, 1e-5, gamma, beta)
        kw = dict(in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
    return x, dy, w, kw


@pytest.mark.parametrize("cin,cout,norm", [(32, 32, True), (4, 32, False), (64, 32, True)])
def test_conv_adjoint_identities_128cube(hip_backend, cin, cout, norm):
    be = hip_backend
    x, dy, w, kw = _conv_triplet(be, 2, cin, cout, norm)
    y = be.empty_act(2, S, S, S, cout)
    be.conv_fwd(x, be.pack_weight(w, 0), y, 3, 1, **kw)
    dw = torch.empty_like(w)
    be.conv_wgrad(x, dy, dw, 3, 1, **kw)
    lhs = _dot(y.tensor(), dy.tensor())
    assert abs(_dot(w, dw) - lhs) / abs(lhs) < 1e-4              # <conv(x;w), dy> == <w, dL/dw>
    if not norm:                                                 # dgrad is the adjoint w.r.t. the (un-normalised) input
        dx = be.empty_act(2, S, S, S, cin)
        be.conv_fwd(dy, be.pack_weight(w, 1), dx, 3, 1)
        assert abs(_dot(x.tensor(), dx.tensor()) - lhs) / abs(lhs) < 1e-4
    # linearity in the weights
    w2 = torch.randn_like(w) * 0.05
    y2, y12 = be.empty_act(2, S, S, S, cout), be.empty_act(2, S, S, S, cout)
    be.conv_fwd(x, be.pack_weight(w2, 0), y2, 3, 1, **kw)
    be.conv_fwd(x, be.pack_weight((w + w2).contiguous(), 0), y12, 3, 1, **kw)
    assert C.rel_err(y12.tensor(), y.tensor() + y2.tensor()) < 1e-5


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 32)])
def test_fp32_mfma_and_split_bf16_kernels_agree_128cube(hip_backend, cin, cout):
    be = hip_backend
    x, dy, w, kw = _conv_triplet(be, 2, cin, cout, True)
    outs = {}
    try:
        for prec in ("fp32", "bf16x6"):
            be.set_precision(prec)
            y = be.empty_act(2, S, S, S, cout)
            dx = be.empty_act(2, S, S, S, cin)
            dw = torch.empty_like(w)
            be.conv_fwd(x, be.pack_weight(w, 0), y, 3, 1, **kw)
            be.conv_fwd(dy, be.pack_weight(w, 1), dx, 3, 1)
            be.conv_wgrad(x, dy, dw, 3, 1, **kw)
            outs[prec] = (y.tensor().clone(), dx.tensor().clone(), dw.clone())
    finally:
        be.set_precision("fp32")
    for a, b in zip(outs["fp32"], outs["bf16x6"]):
        assert C.rel_err(b, a) < 1e-5


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 32)])
def test_winograd_and_direct_kernels_agree_128cube(hip_backend, cin, cout):
    """The product routing (Winograd F(2x2,3x3) x direct-z forward / dgrad, csrc/conv3d_wino.hip) against the direct exact-fp32 MFMA
    kernels (MI355_WINOGRAD=0) on the full-size layers: two independently written kernels, same op."""
    be = hip_backend
    x, dy, w, kw = _conv_triplet(be, 2, cin, cout, True)
    outs = {}
    old = be.winograd
    try:
        for wino in (True, False):
            be.winograd = wino
            y, dx = be.empty_act(2, S, S, S, cout), be.empty_act(2, S, S, S, cin)
            be.conv_fwd(x, be.pack_weight(w, 0), y, 3, 1, **kw)
            be.conv_fwd(dy, be.pack_weight(w, 1), dx, 3, 1)
            outs[wino] = (y.tensor().clone(), dx.tensor().clone())
    finally:
        be.winograd = old
    for a, b in zip(outs[True], outs[False]):
        assert C.rel_err(a, b) < 3e-6
        assert not torch.equal(a, b)                            # they ARE different kernels (the switch took effect)


def test_unet3d_step_128cube_batch2_properties():
    torch.manual_seed(1234)
    m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().eval()
    crit = losses.HipDiceLoss(sigmoid=True)
    x1, y1 = R.synthetic_case(1, 4, (S, S, S), 3)
    x = torch.cat((x1, x1)).cuda()
    y = torch.cat((y1, y1)).cuda()

    def step():
        for p in m.parameters():
            p.grad = None
        out = m(x)
        loss = crit(out, y)
        loss.backward()
        torch.cuda.synchronize()
        return out.detach().clone(), float(loss), [p.grad.detach().clone() for p in m.parameters()]
    o1, l1, g1 = step()
    o2, l2, g2 = step()
    assert torch.isfinite(o1).all() and all(torch.isfinite(g).all() for g in g1)
    assert torch.equal(o1, o2) and l1 == l2 and all(torch.equal(a, b) for a, b in zip(g1, g2))     # deterministic
    assert torch.equal(o1[0], o1[1])                                                               # batch independence
    with torch.no_grad():
        single = m(x[:1])
    assert torch.equal(single[0], o1[0])
    # doubling an identical batch leaves the mean loss unchanged and (sum of two equal halves / 2) the gradient too
    for p in m.parameters():
        p.grad = None
    ls = crit(m(x[:1]), y[:1])
    ls.backward()
    assert abs(float(ls) - l1) / l1 < 1e-6
    for p, g in zip(m.parameters(), g1):
        assert C.rel_err(p.grad, g) < 1e-4


def test_groupnorm_known_statistics_and_perfect_dice_128cube(hip_backend):
    be = hip_backend
    g = torch.Generator(device="cuda").manual_seed(3)
    c = 32
    x = be.empty_act(2, S, S, S, c)
    x.buf.normal_(generator=g)
    mean = torch.linspace(-20, 20, c, device="cuda")
    std = torch.linspace(0.1, 3.0, c, device="cuda")
    t = x.tensor()
    t.sub_(t.mean(dim=(1, 2, 3), keepdim=True)).div_(t.std(dim=(1, 2, 3), keepdim=True, unbiased=False)).mul_(std).add_(mean)
    mr, sc, sh = be.gn_stats(x, c, 0.0, None, None)            # G == C: per-channel (InstanceNorm) statistics
    assert C.rel_err(mr[..., 0], mean.expand(2, c)) < 1e-5
    assert C.rel_err(mr[..., 1], (1.0 / std).expand(2, c)) < 1e-4
    y = (torch.rand(2, 3, S, S, S, device="cuda", generator=g) > 0.7).to(torch.uint8)
    logits = (y.float() * 2 - 1) * 40.0
    loss, _ = be.dice(logits, y)
    assert abs(float(loss)) < 1e-5


def test_sliding_window_partition_of_unity_240x240x155():
    x = torch.randn(1, 3, 240, 240, 155, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    for mode in ("constant", "gaussian"):
        inf = inferer.HipSlidingWindowInferer((128, 128, 128), sw_batch_size=3, overlap=0.5, mode=mode)
        out = inf(x, lambda w: w)
        assert out.shape == x.shape
        assert C.rel_err(out, x) < 1e-6
