"""Per-op parity cases shared by the CPU-emulator tests (test_ops_emu.py, small sizes) and the GPU tests
(test_ops_gpu.py). Each case runs an op through the C ABI (via 3dunetcnn_amd.ops.Backend) and compares with the
torch-CPU oracle (oracle/torch_ops.py). Tolerance: 1e-3 relative (max|a-b| / max|b|), BASELINE.json north_star;
the fp32 MFMA path is normally ~1e-6."""
import importlib
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import torch_ops as O  # noqa: E402

ops = importlib.import_module("3dunetcnn_amd.ops")
Act = ops.Act
TOL = 1e-3


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


KAPPA = 10.0


def grad_parity(grads, g32, g64, floor, tol=TOL, kappa=KAPPA, perturbed=()):
    """Whole-network gradient parity criterion. For each parameter the kernel gradient must EITHER agree with the fp32 CPU
    oracle (the reference's own arithmetic) to `tol`, OR agree with the fp64 evaluation of the same graph to
    max(tol, kappa * floor), floor = the measured response of that gradient to one-ulp (1e-7) perturbations of the fp32
    oracle's convolutions (oracle/conditioning.py). kappa = 10: an fp32 convolution with K = 27*Cin terms carries
    ~0.2*sqrt(K) ulps of roundoff (3 .. 17 ulps for Cin = 32 .. 256), whatever the implementation. The fp32 leg matters when
    the fp64 graph takes a different discrete branch (ReLU masks of exactly-tied values); the fp64 leg when the gradient is
    ill-conditioned and two fp32 evaluations cannot agree. `perturbed` = the one-ulp-perturbed fp32 oracle evaluations the
    floor was measured from: agreeing with any of them to `tol` passes too (a single ReLU tie puts the kernel exactly on the
    branch some of those evaluations take). Returns dict(ratio=worst error/allowance (<= 1 passes), ...)."""
    worst = dict(ratio=0.0, key=None)
    n_ill = n_loose = 0
    max32 = 0.0
    for k, g in grads.items():
        e32, e64 = rel_err(g, g32[k]), rel_err(g, g64[k])
        allow64 = max(tol, kappa * floor[k])
        max32 = max(max32, e32)
        n_ill += kappa * floor[k] > tol
        n_loose += e32 > tol              # tensors that do NOT meet the tolerance against the fp32 oracle directly (they need a loose leg)
        r = min(e32 / tol, e64 / allow64)
        for gp in perturbed:
            if r > 1.0:
                r = min(r, rel_err(g, gp[k]) / tol)
        if r > worst["ratio"]:
            worst = dict(ratio=r, key=k, err_vs_fp32=e32, err_vs_fp64=e64, noise_floor=floor[k])
    worst["n_ill_conditioned"] = n_ill      # a property of the problem (oracle-side conditioning), not of the kernels
    worst["n_loose"] = n_loose
    worst["max_err_vs_fp32"] = max32
    return worst


def assert_recorded(e, bounds):
    """Whole-network parity diagnostics against bounds recorded on MI355X: `grad` (worst error / allowance, <= 1 passes) must keep
    clear of its limit, the number of tensors that need a loose leg (`n_loose`) and the worst error against the fp32 oracle must not
    grow -- so a regression that pushes well-conditioned tensors onto the loose legs fails even though every tensor still 'passes'."""
    for k, b in bounds.items():
        assert e[k] <= b, (k, e[k], "recorded bound", b, e)


def to_act(be, t, ld=None, c0=0):
    """NCDHW cpu tensor -> Act on the backend device with leading dimension ld and channel offset c0."""
    n, c, d, h, w = t.shape
    ld = ld or c
    buf = torch.full((n, d, h, w, ld), 7.0, dtype=torch.float32)  # poison other channels
    buf[..., c0:c0 + c] = t.permute(0, 2, 3, 4, 1)
    return Act(buf.to(be.device).contiguous(), c0, c)


def from_act(a):
    return a.tensor().detach().cpu().permute(0, 4, 1, 2, 3).contiguous()


def dev(be, t):
    return None if t is None else t.detach().clone().to(be.device).contiguous()


def case_conv_fwd(be, n, cin, cout, dhw, kd=3, stride=1, norm=False, groups=None, slope=0.0, residual=False, chscale=False,
                  bias=False, xld=None, yld=None, yc0=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    x = torch.randn(n, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, kd, kd, kd, generator=g) * (1.0 / (cin * kd ** 3) ** 0.5)
    gamma = beta = None
    normspec = None
    if norm:
        groups = groups or (8 if cin >= 8 and cin % 8 == 0 else cin)
        gamma = torch.rand(cin, generator=g) + 0.5
        beta = torch.randn(cin, generator=g) * 0.3
        normspec = (groups, gamma, beta, 1e-5, slope)
    b = torch.randn(cout, generator=g) if bias else None
    pad = kd // 2
    od, oh, ow = [(s + 2 * pad - kd) // stride + 1 for s in dhw]
    res = torch.randn(n, cout, od, oh, ow, generator=g) if residual else None
    cs = (torch.rand(n, cout, generator=g) > 0.3).float() * 1.25 if chscale else None
    ref = O.conv_block(x, wt, stride, pad, normspec, b, res, cs)

    xa = to_act(be, x, xld)
    ya = to_act(be, torch.zeros(n, cout, od, oh, ow), yld, yc0)
    wp = be.pack_weight(dev(be, wt), 0)
    kw = {}
    if norm:
        mr, sc, sh = be.gn_stats(xa, groups, 1e-5, dev(be, gamma), dev(be, beta))
        kw = dict(in_mode=ops.IN_AFFINE_ACT, slope=slope, scale=sc, shift=sh)
    ra = to_act(be, res) if residual else None
    be.conv_fwd(xa, wp, ya, kd, stride, pad, bias=dev(be, b), residual=ra, chscale=dev(be, cs), **kw)
    out = from_act(ya)
    e = rel_err(out, ref)
    # untouched channels of a wider concat buffer must keep the poison value
    if yld and yld > cout:
        other = torch.ones(yld, dtype=torch.bool)
        other[yc0:yc0 + cout] = False
        assert bool((ya.buf.cpu()[..., other] == 7.0).all()), "conv wrote outside its channel slice"
    return e


def case_conv_dgrad(be, n, cin, cout, dhw, stride=1, seed=1, residual=False):
    """dgrad through mi355_conv3d_fwd with the mode-1 pack (stride 1) / zero-insert (stride 2); residual: the skip gradient the network adds in
    the epilogue of the stride-2 data gradient (unet.py: d_skips)."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    x = torch.randn(n, cin, d, h, w, generator=g, requires_grad=True)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (cin * 27) ** 0.5)
    y = F.conv3d(x, wt, None, stride=stride, padding=1)
    dy = torch.randn(y.shape, generator=g)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    dya = to_act(be, dy)
    dxa = to_act(be, torch.zeros(n, cin, d, h, w))
    wpd = be.pack_weight(dev(be, wt), 1)
    res = torch.randn(n, cin, d, h, w, generator=g) if residual else None
    ra = to_act(be, res) if residual else None
    if stride == 1:
        be.conv_fwd(dya, wpd, dxa, 3, 1, 1, residual=ra)
    else:
        be.conv_fwd(dya, wpd, dxa, 3, 1, 1, in_mode=ops.IN_ZERO_INSERT, out_dhw=(d, h, w), residual=ra)
    return rel_err(from_act(dxa), dx_ref + res if residual else dx_ref)


def case_tconv3(be, n, cin, cout, dhw, pad_to=None, seed=12):
    """ConvTranspose3d(k3, s2, p1, bias) forward (decoder.py:99-102) = zero-insert conv with the mode-2 pack; `pad_to`: the
    F.pad window of unet.py:34-40 (output written at offset diff//2 of a pre-zeroed larger tensor)."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    x = torch.randn(n, cin, d, h, w, generator=g)
    wt = torch.randn(cin, cout, 3, 3, 3, generator=g) * (1.0 / (cin * 27 / 8) ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = F.conv_transpose3d(x, wt, b, stride=2, padding=1)
    od = tuple(2 * s - 1 for s in dhw)
    tgt = pad_to or od
    off = tuple((t - o) // 2 for t, o in zip(tgt, od))
    if pad_to:
        ref = F.pad(ref, [v for t, o, f in reversed(list(zip(tgt, od, off))) for v in (f, t - o - f)])
    xa = to_act(be, x)
    ya = to_act(be, torch.zeros(n, cout, *tgt))
    be.conv_fwd(xa, be.pack_weight(dev(be, wt), 2), ya, 3, 1, pad=1, in_mode=ops.IN_ZERO_INSERT, bias=dev(be, b), off=off, out_dhw=od)
    return rel_err(from_act(ya), ref)


def case_conv_wgrad(be, n, cin, cout, dhw, kd=3, stride=1, norm=False, slope=0.0, seed=2):
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    x = torch.randn(n, cin, d, h, w, generator=g)
    wt = (torch.randn(cout, cin, kd, kd, kd, generator=g) * 0.1).requires_grad_(True)
    normspec = None
    gamma = beta = None
    groups = 8 if cin >= 8 and cin % 8 == 0 else cin
    if norm:
        gamma = torch.rand(cin, generator=g) + 0.5
        beta = torch.randn(cin, generator=g) * 0.3
        normspec = (groups, gamma, beta, 1e-5, slope)
    pad = kd // 2
    y = O.conv_block(x, wt, stride, pad, normspec)
    dy = torch.randn(y.shape, generator=g)
    (dw_ref,) = torch.autograd.grad(y, wt, dy)
    xa, dya = to_act(be, x), to_act(be, dy)
    dw = torch.full(wt.shape, 3.0, dtype=torch.float32, device=be.device)
    kw = {}
    if norm:
        mr, sc, sh = be.gn_stats(xa, groups, 1e-5, dev(be, gamma), dev(be, beta))
        kw = dict(in_mode=ops.IN_AFFINE_ACT, slope=slope, scale=sc, shift=sh)
    be.conv_wgrad(xa, dya, dw, kd, stride, pad, **kw)
    return rel_err(dw, dw_ref)


def case_tconv2(be, n, cin, cout, dhw, norm=True, yld=None, seed=9):
    """ConvTranspose3d(k2, s2, bias=False) (MONAI UnetUpBlock.transp_conv) as one 1x1x1 GEMM with a depth-to-space epilogue,
    its dgrad through the space-to-depth prologue and its wgrad; input = LeakyReLU(InstanceNorm(x)) applied in the prologue."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    x = torch.randn(n, cin, d, h, w, generator=g).requires_grad_(True)
    wt = (torch.randn(cin, cout, 2, 2, 2, generator=g) * (1.0 / cin ** 0.5)).requires_grad_(True)
    gamma = torch.rand(cin, generator=g) + 0.5
    beta = torch.randn(cin, generator=g) * 0.3
    a = O.norm_act(x, cin, gamma, beta, 1e-5, 0.01) if norm else x
    a.retain_grad()
    y = F.conv_transpose3d(a, wt, None, stride=2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xa = to_act(be, x.detach())
    kw = {}
    if norm:
        mr, sc, sh = be.gn_stats(xa, cin, 1e-5, dev(be, gamma), dev(be, beta))
        kw = dict(in_mode=ops.IN_AFFINE_ACT, slope=0.01, scale=sc, shift=sh)
    w1 = dev(be, wt.detach().permute(2, 3, 4, 1, 0).reshape(8 * cout, cin, 1, 1, 1))
    ya = to_act(be, torch.zeros(n, cout, 2 * d, 2 * h, 2 * w), yld)
    be.conv_fwd(xa, be.pack_weight(w1, 0), ya, 1, out_mode=ops.OUT_D2S, **kw)
    e_f = rel_err(from_act(ya), y)
    dya = to_act(be, dy, yld)
    dxa = to_act(be, torch.zeros(n, cin, d, h, w))
    be.conv_fwd(dya, be.pack_weight(w1, 1), dxa, 1, in_mode=ops.IN_S2D)
    e_d = rel_err(from_act(dxa), a.grad)
    dw1 = torch.full((8 * cout, cin), 3.0, dtype=torch.float32, device=be.device)
    be.conv_wgrad(xa, dya, dw1, 1, out_mode=ops.OUT_D2S, **kw)
    dw = dw1.view(2, 2, 2, cout, cin).permute(4, 3, 0, 1, 2)
    return dict(fwd=e_f, dgrad=e_d, wgrad=rel_err(dw, wt.grad))


def case_conv_cat_slope(be, n, c_up, c_skip, cout, dhw, stride=1, seed=10):
    """Conv over a concat buffer whose first c_up channels are raw (identity prologue) and whose last c_skip channels get
    InstanceNorm + LeakyReLU(0.01) in the prologue (per-channel slope): fwd and wgrad."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    up = torch.randn(n, c_up, d, h, w, generator=g)
    sk = torch.randn(n, c_skip, d, h, w, generator=g) * 1.5 + 0.3
    gamma = torch.rand(c_skip, generator=g) + 0.5
    beta = torch.randn(c_skip, generator=g) * 0.3
    wt = (torch.randn(cout, c_up + c_skip, 3, 3, 3, generator=g) * 0.05).requires_grad_(True)
    a = torch.cat((up, O.norm_act(sk, c_skip, gamma, beta, 1e-5, 0.01)), 1)
    y = F.conv3d(a, wt, None, stride=stride, padding=1)
    dy = torch.randn(y.shape, generator=g)
    (dw_ref,) = torch.autograd.grad(y, wt, dy)
    cat = to_act(be, torch.cat((up, sk), 1))
    mr, sc, sh = be.gn_stats(cat.slice(c_up, c_skip), c_skip, 1e-5, dev(be, gamma), dev(be, beta))
    ones = torch.ones(n, c_up, device=be.device)
    scale = torch.cat((ones, sc), 1).contiguous()
    shift = torch.cat((0 * ones, sh), 1).contiguous()
    slope = torch.cat((ones[0], torch.full((c_skip,), 0.01, device=be.device))).contiguous()
    kw = dict(in_mode=ops.IN_AFFINE_ACT, slope=0.01, scale=scale, shift=shift, in_slope=slope)
    ya = to_act(be, torch.zeros_like(y.detach()))
    be.conv_fwd(cat, be.pack_weight(dev(be, wt.detach()), 0), ya, 3, stride, **kw)
    dw = torch.empty(wt.shape, dtype=torch.float32, device=be.device)
    be.conv_wgrad(cat, to_act(be, dy), dw, 3, stride, **kw)
    return dict(fwd=rel_err(from_act(ya), y), wgrad=rel_err(dw, dw_ref))


def case_gn(be, n, c, dhw, groups, slope=0.0, ld=None, seed=3, offset=0.4):
    """stats + act(GN) backward (incl. addend) against autograd of F.group_norm -> (leaky)relu (double precision truth).
    offset / 1.7 is the mean/std ratio of the input: large ratios break single-pass E[x^2]-E[x]^2 statistics."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    x = (torch.randn(n, c, d, h, w, generator=g) * 1.7 + offset).double().requires_grad_(True)
    gamma = (torch.rand(c, generator=g) + 0.5).double().requires_grad_(True)
    beta = (torch.randn(c, generator=g) * 0.3).double().requires_grad_(True)
    a = O.norm_act(x, groups, gamma, beta, 1e-5, slope)
    dA = torch.randn(a.shape, generator=g)
    add = torch.randn(a.shape, generator=g)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(a, (x, gamma, beta), dA.double())
    dx_ref = dx_ref + add
    x, gamma, beta = x.detach().float(), gamma.detach().float(), beta.detach().float()
    xa = to_act(be, x.detach(), ld)
    mr, sc, sh = be.gn_stats(xa, groups, 1e-5, dev(be, gamma.detach()), dev(be, beta.detach()))
    xd = x.detach()
    xg = xd.reshape(n, groups, -1)
    mean_ref = xg.mean(-1)
    rstd_ref = (xg.var(-1, unbiased=False) + 1e-5).rsqrt()
    e_stats = max(rel_err(mr[..., 0], mean_ref), rel_err(mr[..., 1], rstd_ref))
    dAa = to_act(be, dA)
    dxa = to_act(be, torch.zeros_like(dA))
    dgam = torch.empty(c, device=be.device)
    dbet = torch.empty(c, device=be.device)
    be.gn_act_bwd(xa, dAa, dxa, groups, slope, dev(be, gamma.detach()), mr, sc, sh, dgam, dbet, addend=to_act(be, add))
    return dict(stats=e_stats, dx=rel_err(from_act(dxa), dx_ref), dgamma=rel_err(dgam, dg_ref), dbeta=rel_err(dbet, db_ref))


def case_conv_moments(be, n, cin, cout, dhw, stride=1, norm=True, residual=False, chscale=False, groups_out=None, yld=None, yc0=0,
                      expect_fused=True, seed=13, ytol=TOL, strict_vs_oracle=True):
    """Norm statistics of a conv OUTPUT taken from the moment records its epilogue wrote (csrc/gn_fuse.h; myronenko.py:17-21:
    every conv output is the next block's GroupNorm input) against the statistics of the oracle's conv output, and against the
    standalone statistics pass over the same device tensor. Returns the worst relative error of (mean, rstd, scale, shift)."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    x = torch.randn(n, cin, d, h, w, generator=g) * 1.3 + 0.2
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (cin * 27) ** 0.5)
    normspec = None
    kw = {}
    xa = to_act(be, x)
    if norm:
        gi = 8 if cin >= 8 and cin % 8 == 0 else cin
        gamma = torch.rand(cin, generator=g) + 0.5
        beta = torch.randn(cin, generator=g) * 0.3
        normspec = (gi, gamma, beta, 1e-5, 0.0)
        mr, sc, sh = be.gn_stats(xa, gi, 1e-5, dev(be, gamma), dev(be, beta))
        kw = dict(in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
    od, oh, ow = [(s + 2 - 3) // stride + 1 for s in dhw]
    res = torch.randn(n, cout, od, oh, ow, generator=g) + 3.0 if residual else None      # |mean| >> std of the conv term
    cs = (torch.rand(n, cout, generator=g) > 0.3).float() * 1.25 if chscale else None
    ref = O.conv_block(x, wt, stride, 1, normspec, None, res, cs)
    ya = to_act(be, torch.zeros(n, cout, od, oh, ow), yld, yc0)
    be.conv_fwd(xa, be.pack_weight(dev(be, wt), 0), ya, 3, stride, 1, residual=to_act(be, res) if residual else None, chscale=dev(be, cs),
                moments=True, **kw)
    assert rel_err(from_act(ya), ref) < ytol
    assert (ya.mom is not None) == expect_fused, "fused statistics expected" if expect_fused else "fallback expected"
    go = groups_out or (8 if cout >= 8 and cout % 8 == 0 else cout)
    g2 = torch.rand(cout, generator=g) + 0.5
    b2 = torch.randn(cout, generator=g) * 0.3
    mr, sc, sh = be.gn_stats(ya, go, 1e-5, dev(be, g2), dev(be, b2))
    rg = ref.double().reshape(n, go, -1)
    mean_ref, rstd_ref = rg.mean(-1), (rg.var(-1, unbiased=False) + 1e-5).rsqrt()
    cpg = cout // go
    sc_ref = g2.double()[None] * rstd_ref.repeat_interleave(cpg, 1)
    sh_ref = b2.double()[None] - mean_ref.repeat_interleave(cpg, 1) * sc_ref
    e = max(rel_err(mr[..., 0], mean_ref), rel_err(mr[..., 1], rstd_ref), rel_err(sc, sc_ref), rel_err(sh, sh_ref))
    if not strict_vs_oracle:      # reduced-precision conv arithmetic: the statistics are those of the stored tensor, checked below
        assert e < 10 * ytol
        e = 0.0
    # cross-check: the standalone pass over the same device tensor
    saved, ya.mom = ya.mom, None
    mr2, sc2, sh2 = be.gn_stats(ya, go, 1e-5, dev(be, g2), dev(be, b2))
    ya.mom = saved
    e = max(e, rel_err(mr[..., 1], mr2[..., 1].cpu()), rel_err(sh, sh2.cpu()))
    return e


def case_cat_moments(be, n, c_up, c_skip, dhw, seed=14):
    """Statistics of a concat buffer whose two halves have different producers (segmentation/unet.py:42): the skip half's records come
    from a conv epilogue, the up half's from the standalone record producer (Backend.moments); groups straddle nothing or both."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    up = torch.randn(n, c_up, d, h, w, generator=g) * 0.7 - 1.0
    x = torch.randn(n, 8, d, h, w, generator=g)
    wt = torch.randn(c_skip, 8, 3, 3, 3, generator=g) * 0.1
    skip = F.conv3d(x, wt, None, padding=1)
    cat = to_act(be, torch.zeros(n, c_up + c_skip, d, h, w))
    cat.tensor()[..., :c_up] = up.permute(0, 2, 3, 4, 1).to(be.device)
    sl_up, sl_skip = cat.slice(0, c_up), cat.slice(c_up, c_skip)
    be.conv_fwd(to_act(be, x), be.pack_weight(dev(be, wt), 0), sl_skip, 3, 1, 1, moments=True)
    assert sl_skip.mom is not None
    be.moments(sl_up)
    cat.mom = sl_up.mom + sl_skip.mom
    c = c_up + c_skip
    groups = 8 if c % 8 == 0 else c
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    mr, sc, sh = be.gn_stats(cat, groups, 1e-5, dev(be, gamma), dev(be, beta))
    ref = torch.cat((up, skip), 1).double().reshape(n, groups, -1)
    mean_ref, rstd_ref = ref.mean(-1), (ref.var(-1, unbiased=False) + 1e-5).rsqrt()
    return max(rel_err(mr[..., 0], mean_ref), rel_err(mr[..., 1], rstd_ref))


def case_gn_bwd_fused(be, n, cin, cout, dhw, groups=None, slope=0.0, expect_fused=True, seed=15, compare_unfused=False):
    """act(GroupNorm) backward where the first pass (sum du, sum du*xhat) leaves with the epilogue of the dgrad conv that produces
    dA (csrc/gn_fuse.h): dx / dgamma / dbeta against double-precision autograd of conv(act(norm(x)))."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    groups = groups or (8 if cin >= 8 and cin % 8 == 0 else cin)
    x = (torch.randn(n, cin, d, h, w, generator=g) * 1.7 + 0.4).double().requires_grad_(True)
    gamma = (torch.rand(cin, generator=g) + 0.5).double().requires_grad_(True)
    beta = (torch.randn(cin, generator=g) * 0.3).double().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (cin * 27) ** 0.5)).double()
    y = F.conv3d(O.norm_act(x, groups, gamma, beta, 1e-5, slope), wt, None, padding=1)
    dy = torch.randn(y.shape, generator=g)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(y, (x, gamma, beta), dy.double())
    xa = to_act(be, x.detach().float())
    st = be.gn_stats(xa, groups, 1e-5, dev(be, gamma.detach().float()), dev(be, beta.detach().float()))
    dA = to_act(be, torch.zeros(n, cin, d, h, w))
    parts = be.conv_fwd(to_act(be, dy), be.pack_weight(dev(be, wt.float()), 1), dA, 3, 1, 1, gnb=(xa, st, groups, slope))
    assert (parts is not None) == expect_fused
    dgam, dbet = torch.empty(cin, device=be.device), torch.empty(cin, device=be.device)
    dA_raw = dA.tensor().clone()
    be.gn_act_bwd(xa, dA, dA, groups, slope, dev(be, gamma.detach().float()), st[0], st[1], st[2], dgam, dbet, partials=parts)
    if compare_unfused:
        # reduced-precision conv arithmetic: compare with the unfused norm backward applied to the SAME dA (the sums' own accuracy)
        dA2 = Act(dA_raw.contiguous(), 0, cin)
        dg2, db2 = torch.empty(cin, device=be.device), torch.empty(cin, device=be.device)
        be.gn_act_bwd(xa, dA2, dA2, groups, slope, dev(be, gamma.detach().float()), st[0], st[1], st[2], dg2, db2)
        return dict(dx=rel_err(from_act(dA), from_act(dA2)), dgamma=rel_err(dgam, dg2.cpu()), dbeta=rel_err(dbet, db2.cpu()))
    return dict(dx=rel_err(from_act(dA), dx_ref), dgamma=rel_err(dgam, dg_ref), dbeta=rel_err(dbet, db_ref))


def case_c4_bwd(be, n, dhw, groups=4, slope=0.0, seed=21, xld=None, dyld=None):
    """Fused backward of the network's first pair [GroupNorm(4) -> act -> Conv3d(4 -> 32, k3)] (csrc/conv3d_c4_bwd.hip): dW of the conv
    and dgamma / dbeta of the norm in one pass over dy, against autograd in double precision."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    x = (torch.randn(n, 4, d, h, w, generator=g) * 1.3 + 0.2).double()
    gamma = (torch.rand(4, generator=g) + 0.5).double().requires_grad_(True)
    beta = (torch.randn(4, generator=g) * 0.3).double().requires_grad_(True)
    wt = (torch.randn(32, 4, 3, 3, 3, generator=g) * 0.1).double().requires_grad_(True)
    y = F.conv3d(O.norm_act(x, groups, gamma, beta, 1e-5, slope), wt, padding=1)
    dy = torch.randn(y.shape, generator=g)
    dw_ref, dg_ref, db_ref = torch.autograd.grad(y, (wt, gamma, beta), dy.double())
    xa, dya = to_act(be, x.float(), xld), to_act(be, dy, dyld)
    gam, bet = dev(be, gamma.detach().float()), dev(be, beta.detach().float())
    mr, sc, sh = be.gn_stats(xa, groups, 1e-5, gam, bet)
    assert be.c4_bwd_supported(xa, dya, ops.IN_AFFINE_ACT, slope, sc, sh)
    wp = be.pack_weight(dev(be, wt.detach().float()), 1)
    dw = torch.full((32, 4, 3, 3, 3), 7.0, device=be.device)
    dgam, dbet = torch.full((4,), 7.0, device=be.device), torch.full((4,), 7.0, device=be.device)
    be.c4_bwd(xa, dya, wp, dw, groups, gam, mr, sc, sh, dgam, dbet, slope=slope)
    return dict(dw=rel_err(dw, dw_ref), dgamma=rel_err(dgam, dg_ref), dbeta=rel_err(dbet, db_ref))


def case_upsample(be, n, c, lo_dhw, target_dhw, c_skip=8, seed=4):
    g = torch.Generator().manual_seed(seed)
    lo = torch.randn(n, c, *lo_dhw, generator=g, requires_grad=True)
    up = O.upsample_pad(lo, target_dhw)
    dcat = torch.randn(n, c + c_skip, *target_dhw, generator=g)
    (dlo_ref,) = torch.autograd.grad(up, lo, dcat[:, :c])
    off = tuple((t - 2 * l) // 2 for t, l in zip(target_dhw, lo_dhw))
    loa = to_act(be, lo.detach())
    cat = to_act(be, torch.zeros(n, c, *target_dhw), ld=c + c_skip, c0=0)
    be.upsample2x_fwd(loa, cat, off)
    e_f = rel_err(from_act(cat), up)
    dcata = to_act(be, dcat).slice(0, c)
    dloa = to_act(be, torch.zeros_like(lo.detach()))
    be.upsample2x_bwd(dcata, dloa, off)
    return dict(fwd=e_f, bwd=rel_err(from_act(dloa), dlo_ref))


def case_proj(be, n, cin, cout, dhw, bias=False, seed=5, norm=False):
    """norm=True: the projection reads LeakyReLU(InstanceNorm(x)) through its prologue (DynUNet output block); dx is then
    the gradient wrt the activated input."""
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(n, cin, *dhw, generator=g)
    pk = {}
    if norm:
        gamma = torch.rand(cin, generator=g) + 0.5
        beta = torch.randn(cin, generator=g) * 0.3
        x = O.norm_act(x0, cin, gamma, beta, 1e-5, 0.01).detach().requires_grad_(True)
    else:
        x = x0.clone().requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, 1, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True) if bias else None
    y = F.conv3d(x, w, b)
    dy = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad(y, (x, w) + ((b,) if bias else ()), dy)
    xa = to_act(be, x0)
    if norm:
        mr, sc, sh = be.gn_stats(xa, cin, 1e-5, dev(be, gamma), dev(be, beta))
        pk = dict(scale=sc, shift=sh, slope=0.01)
    logits = torch.empty(n, cout, *dhw, device=be.device)
    be.proj_fwd(xa, dev(be, w.detach().reshape(cout, cin)), dev(be, b.detach()) if bias else None, logits, **pk)
    dxa = to_act(be, torch.zeros_like(x.detach()))
    dw = torch.empty(cout, cin, device=be.device)
    dbias = torch.empty(cout, device=be.device) if bias else None
    be.proj_bwd(xa, dev(be, w.detach().reshape(cout, cin)), dev(be, dy), dxa, dw, dbias, **pk)
    out = dict(fwd=rel_err(logits, y), dx=rel_err(from_act(dxa), grads[0]), dw=rel_err(dw, grads[1].reshape(cout, cin)))
    if bias:
        out["dbias"] = rel_err(dbias, grads[2])
    return out


def nested_masks(n, dhw, seed=0):
    """BraTS-like nested binary masks WT >= TC >= ET: three concentric ellipsoids, uint8 [n,3,d,h,w]."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dhw
    zz, yy, xx = torch.meshgrid(torch.arange(d), torch.arange(h), torch.arange(w), indexing="ij")
    out = torch.zeros(n, 3, d, h, w, dtype=torch.uint8)
    for i in range(n):
        cz, cy, cx = [s / 2 + float(torch.rand(1, generator=g) * 2 - 1) * min(8, s / 8) for s in dhw]
        for k, frac in enumerate((0.30, 0.20, 0.10)):
            r = ((zz - cz) / (frac * d)) ** 2 + ((yy - cy) / (frac * h)) ** 2 + ((xx - cx) / (frac * w)) ** 2
            out[i, k] = (r <= 1.0).to(torch.uint8)
    return out


def case_dice(be, n, c, dhw, batch=False, squared=False, u8=True, seed=6):
    g = torch.Generator().manual_seed(seed)
    z = (torch.randn(n, c, *dhw, generator=g) * 2).requires_grad_(True)
    t = nested_masks(n, dhw, seed)[:, :c] if c <= 3 else (torch.rand(n, c, *dhw, generator=g) > 0.7).to(torch.uint8)
    tt = t if u8 else t.float()
    ref = O.dice_loss(z, t, True, batch, squared)
    (dz_ref,) = torch.autograd.grad(ref, z)
    loss, dz = be.dice(dev(be, z.detach()), dev(be, tt), True, batch, squared)
    return dict(loss=abs(float(loss.cpu()) - float(ref.detach())) / abs(float(ref.detach())), grad=rel_err(dz, dz_ref))


def case_ce(be, n, c, dhw, mode="softmax", u8=True, with_dice=False, seed=11):
    """mi355_ce_fwd_bwd vs torch: CrossEntropyLoss(mean) on probability targets / BCEWithLogitsLoss(mean); with_dice: the value
    and gradient accumulate onto a Dice term (0.7 * Dice + 0.4 * CE), as HipDiceCELoss runs it."""
    g = torch.Generator().manual_seed(seed)
    z = (torch.randn(n, c, *dhw, generator=g) * 3).requires_grad_(True)
    t = nested_masks(n, dhw, seed)[:, :c] if c <= 3 else (torch.rand(n, c, *dhw, generator=g) > 0.7).to(torch.uint8)
    tt = t if u8 else t.float()
    ce = F.cross_entropy(z, t.float()) if mode == "softmax" else F.binary_cross_entropy_with_logits(z, t.float())
    ref = 0.4 * ce + (0.7 * O.dice_loss(z, t, True) if with_dice else 0.0)
    (dz_ref,) = torch.autograd.grad(ref, z)
    zd, td = dev(be, z.detach()), dev(be, tt)
    loss = dz = None
    if with_dice:
        loss, dz = be.dice(zd, td, True, grad_scale=0.7)
        loss.mul_(0.7)
    loss, dz = be.cross_entropy(zd, td, mode=mode, weight=0.4, loss=loss, dlogits=dz)
    return dict(loss=abs(float(loss.cpu()) - float(ref.detach())) / abs(float(ref.detach())), grad=rel_err(dz, dz_ref))


def case_adam(be, count, steps=3, wd=0.0, seed=7):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(count, generator=g)
    m = torch.zeros(count)
    v = torch.zeros(count)
    pd, md, vd = dev(be, p), dev(be, m), dev(be, v)
    pr = p.clone()
    for s in range(1, steps + 1):
        gr = torch.randn(count, generator=g) * 0.1
        O.adam_step(pr, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, wd, s)
        be.adam_step(pd, dev(be, gr), md, vd, 1e-3, 0.9, 0.999, 1e-8, wd, s)
    return rel_err(pd, pr)


def case_layout(be, n, c, dhw, seed=8):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, *dhw, generator=g)
    a = to_act(be, torch.zeros_like(x))
    be.ncdhw_to_ndhwc(dev(be, x), a)
    e1 = rel_err(from_act(a), x)
    back = torch.empty_like(x, device=be.device)
    be.ndhwc_to_ncdhw(a, back)
    return max(e1, rel_err(back, x))
