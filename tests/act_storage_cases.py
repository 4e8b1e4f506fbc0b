"""TEST INFRASTRUCTURE: op-level cases of 16-bit activation storage (mi355_act.dtype = MI355_ACT_BF16, csrc/act_io.h), shared by the
emulator tests (tests/test_act_storage_emu.py) and their GPU twins (tests/test_act_storage_gpu.py).

The property checked: a kernel on bf16 tensors computes in fp32 exactly what it computes on fp32 tensors holding the same values, and
rounds ONCE, to nearest even, when it stores. So every case runs an op twice -- on bf16 Acts, and on fp32 Acts filled with the same
(bf16-representable) values -- and demands stored == round_bf16(fp32 result): `launch_audit.store_err`, whose half-ulp allowance is the
storage format and whose remainder is held to 1e-6 (the two runs share code and accumulation order; a value may land on the other side of a
rounding boundary only through instruction-level differences such as fused multiply-adds). Reference semantics: the conv outputs of the
reference's AutocastUNet are 16-bit tensors under torch autocast (unet3d/models/pytorch/segmentation/unet.py:53-58).
Fused statistics are checked against the standalone statistics of the tensor AS STORED (gn_fuse.h records are taken over stored values).
"""
import importlib

import torch

import launch_audit as A
import op_cases as C

ops = importlib.import_module("3dunetcnn_amd.ops")
BF = torch.bfloat16          # the 16-bit storage type under test: `storage_type(torch.float16)` switches the cases to MI355_ACT_F16
TOL = 1e-6


class storage_type:
    """with storage_type(torch.float16): ... -- runs the cases on IEEE fp16 tensors (the reference's own amp tensors, train/train.py:33-37);
    the backend precision must be the matching one ("fp16")."""

    def __init__(self, dt):
        self.dt = dt

    def __enter__(self):
        global BF
        self.saved, BF = BF, self.dt

    def __exit__(self, *a):
        global BF
        BF = self.saved


def bf16_values(shape, g, scale=1.0, shift=0.0):
    return ((torch.randn(shape, generator=g) * scale + shift).to(BF)).float()


def acts(be, t, ld=None, c0=0):
    """NCDHW fp32 tensor of bf16-representable values -> (fp32 Act, bf16 Act) holding them (poisoned other channels)."""
    a32 = C.to_act(be, t, ld, c0)
    a16 = ops.Act(a32.buf.to(BF).contiguous(), a32.c0, a32.c)
    assert torch.equal(a16.tensor().float(), a32.tensor())
    return a32, a16


def out_pair(be, shape5, ld=None, c0=0, fill=0.0):
    n, c, d, h, w = shape5
    t = torch.full((n, c, d, h, w), fill)
    return acts(be, t, ld, c0)


def stored_ok(a16, a32):
    return A.store_err(C.from_act(a16), C.from_act(a32), BF)


def case_cast(be):
    g = torch.Generator().manual_seed(0)
    t = torch.randn(2, 8, 3, 5, 7, generator=g) * 5.0
    x = C.to_act(be, t, ld=12, c0=4)
    y = be.cast(x, BF)
    assert y.dtype == BF and torch.equal(y.tensor().cpu(), x.tensor().cpu().to(BF))          # round to nearest even, as torch
    z = be.cast(y, torch.float32)
    assert z.dtype == torch.float32 and torch.equal(z.tensor().cpu(), y.tensor().cpu().float())
    w = be.cast(y, BF)                                                                       # same type: a copy
    assert torch.equal(w.tensor().cpu(), y.tensor().cpu())
    return 0.0


def case_pointwise(be):
    g = torch.Generator().manual_seed(1)
    errs = {}
    lo32, lo16 = acts(be, bf16_values((2, 8, 3, 4, 5), g))
    cat32, cat16 = out_pair(be, (2, 8, 7, 8, 11), ld=16, c0=8)
    for lo, cat in ((lo32, cat32), (lo16, cat16)):
        be.upsample2x_fwd(lo, cat, (0, 0, 1))
    errs["upsample_fwd"] = stored_ok(cat16, cat32)
    dc32, dc16 = acts(be, bf16_values((2, 8, 7, 8, 11), g))
    dl32, dl16 = out_pair(be, (2, 8, 3, 4, 5))
    for dc, dl in ((dc32, dl32), (dc16, dl16)):
        be.upsample2x_bwd(dc, dl, (0, 0, 1))
    errs["upsample_bwd"] = stored_ok(dl16, dl32)
    a32, a16 = acts(be, bf16_values((1, 12, 3, 3, 5), g))
    b32, b16 = acts(be, bf16_values((1, 12, 3, 3, 5), g), ld=16)
    y32, y16 = out_pair(be, (1, 12, 3, 3, 5))
    be.add(a32, b32, y32); be.add(a16, b16, y16)
    errs["add"] = stored_ok(y16, y32)
    s = C.dev(be, torch.rand(1, 12, generator=g) + 0.5)
    be.chscale(a32, s, y32); be.chscale(a16, s, y16)
    errs["chscale"] = stored_ok(y16, y32)
    # layout changes at the module boundary
    src = torch.randn(2, 4, 3, 4, 5, generator=g)
    d16 = be.empty_act(2, 3, 4, 5, 4, dtype=BF)
    be.ncdhw_to_ndhwc(C.dev(be, src), d16)
    assert torch.equal(d16.tensor().cpu().permute(0, 4, 1, 2, 3), src.to(BF))
    back = torch.empty(2, 4, 3, 4, 5, device=be.device)
    be.ndhwc_to_ncdhw(d16, back)
    assert torch.equal(back.cpu(), src.to(BF).float())
    return errs


def case_norm(be, n=2, c=16, dhw=(5, 6, 7), groups=8, slope=0.0):
    g = torch.Generator().manual_seed(2)
    x32, x16 = acts(be, bf16_values((n, c, *dhw), g, 1.7, 0.4), ld=c + 8, c0=8)
    gamma = C.dev(be, torch.rand(c, generator=g) + 0.5)
    beta = C.dev(be, torch.randn(c, generator=g) * 0.3)
    st32 = be.gn_stats(x32, groups, 1e-5, gamma, beta)
    st16 = be.gn_stats(x16, groups, 1e-5, gamma, beta)
    errs = {"stats": max(C.rel_err(a, b) for a, b in zip(st16, st32))}           # same values in: same statistics (fp32 outputs)
    dA32, dA16 = acts(be, bf16_values((n, c, *dhw), g))
    ad32, ad16 = acts(be, bf16_values((n, c, *dhw), g))
    dx32, dx16 = out_pair(be, (n, c, *dhw))
    gr = []
    for x, dA, ad, dx, st in ((x32, dA32, ad32, dx32, st32), (x16, dA16, ad16, dx16, st16)):
        dg, db = torch.empty(c, device=be.device), torch.empty(c, device=be.device)
        be.gn_act_bwd(x, dA, dx, groups, slope, gamma, st[0], st[1], st[2], dg, db, addend=ad)
        gr.append((dg.cpu(), db.cpu()))
    errs["dx"] = stored_ok(dx16, dx32)
    errs["dgamma"] = C.rel_err(gr[1][0], gr[0][0])
    errs["dbeta"] = C.rel_err(gr[1][1], gr[0][1])
    return errs


def case_proj(be):
    g = torch.Generator().manual_seed(3)
    x32, x16 = acts(be, bf16_values((2, 32, 4, 5, 6), g))
    w = C.dev(be, torch.randn(3, 32, generator=g) * 0.2)
    lo32 = torch.empty(2, 3, 4, 5, 6, device=be.device); lo16 = torch.empty_like(lo32)
    be.proj_fwd(x32, w, None, lo32); be.proj_fwd(x16, w, None, lo16)
    errs = {"logits": C.rel_err(lo16, lo32)}
    dz = C.dev(be, torch.randn(2, 3, 4, 5, 6, generator=g))
    dx32, dx16 = out_pair(be, (2, 32, 4, 5, 6))
    dws = []
    for x, dx in ((x32, dx32), (x16, dx16)):
        dw = torch.empty(3, 32, device=be.device)
        be.proj_bwd(x, w, dz, dx, dw, None)
        dws.append(dw.cpu())
    errs["dx"] = stored_ok(dx16, dx32)
    errs["dw"] = C.rel_err(dws[1], dws[0])
    return errs


def _conv_both(be, t, w, mode, kd, yshape, call_kw, x_ld=None, x_c0=0, res=None, gnb_of=None, y_ld=None, y_c0=0, zero=False, x16_is_f32=False, y16_is_f32=False):
    """Runs conv_fwd on (fp32 acts) and (bf16 acts); returns dict(y=stored error, + statistics checks)."""
    if x16_is_f32:
        x32 = x16 = C.to_act(be, t, x_ld, x_c0)             # an fp32 input in both runs (the network's input volume)
    else:
        x32, x16 = acts(be, t, x_ld, x_c0)
    wp = be.pack_weight(C.dev(be, w), mode)
    out = {}
    ys, extra = [], []
    for kind, x in (("f32", x32), ("b16", x16)):
        ydt = torch.float32 if (kind == "f32" or y16_is_f32) else BF
        n, c, d, h, wd = yshape
        ld = y_ld or c
        buf = (torch.zeros if zero else torch.empty)(n, d, h, wd, ld, dtype=ydt, device=be.device)
        if not zero:
            buf.fill_(3.0)
        y = ops.Act(buf, y_c0, c)
        kw = dict(call_kw)
        if res is not None:
            r32, r16 = res
            kw["residual"] = r32 if ydt == torch.float32 else r16
        if gnb_of is not None:
            gx32, gx16, st, groups = gnb_of
            kw["gnb"] = (gx32 if ydt == torch.float32 else gx16, st, groups, 0.0)
        ret = be.conv_fwd(x, wp, y, kd, **kw)
        ys.append(y)
        extra.append((y.mom, ret))
    out["y"] = stored_ok(ys[1], ys[0])
    y16 = ys[1]
    mom, ret = extra[1]
    if call_kw.get("moments"):
        assert mom is not None, "the epilogue did not leave the moment records"
        # fused statistics of the bf16 output == standalone statistics of the tensor as stored
        groups = 8 if y16.c % 8 == 0 else y16.c
        fused = be.gn_stats(y16, groups, 1e-5, None, None)
        y16.mom = None
        alone = be.gn_stats(y16, groups, 1e-5, None, None)
        out["moments"] = max(C.rel_err(a, b) for a, b in zip(fused, alone))
    if gnb_of is not None:
        out["gnb_fused"] = ret is not None
        if ret is not None:
            gx32, gx16, st, groups = gnb_of
            c = y16.c
            res_ = []
            for partials in (ret, None):
                dg, db = torch.empty(c, device=be.device), torch.empty(c, device=be.device)
                dx = be.empty_act(*y16.shape, dtype=BF)
                be.gn_act_bwd(gx16, y16, dx, groups, 0.0, None, st[0], st[1], st[2], dg, db, partials=partials)
                res_.append((dg.cpu(), db.cpu()))
            # the epilogue's sums (taken over the gradient AS STORED) against the standalone first pass over the stored tensor
            out["gnb"] = max(C.rel_err(res_[0][0], res_[1][0]), C.rel_err(res_[0][1], res_[1][1]))
    return out


def case_conv_k1(be, cin=64, cout=32, residual=False, dhw=(4, 5, 7)):
    """(32, 64), (64, 32), (64, 64) channels on bf16 tensors: conv3d_k1_stream_bf16; residual: the gradient accumulation of the data-gradient calls."""
    g = torch.Generator().manual_seed(4)
    t = bf16_values((2, cin, *dhw), g)
    w = torch.randn(cout, cin, 1, 1, 1, generator=g) * 0.1
    res = acts(be, bf16_values((2, cout, *dhw), g)) if residual else None
    return _conv_both(be, t, w, 0, 1, (2, cout, *dhw), {}, x_ld=cin + 8, x_c0=8, y_ld=cout + 32, y_c0=32, res=res)


def case_conv_s2(be, cin=32, cout=64, moments=True):
    g = torch.Generator().manual_seed(5)
    t = bf16_values((2, cin, 9, 10, 13), g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05
    return _conv_both(be, t, w, 0, 3, (2, cout, 5, 5, 7), dict(stride=2, moments=moments))


def case_conv_zero_insert(be, cin=64, cout=32, window=False):
    """dgrad of a stride-2 conv (residual = the skip gradient) / ConvTranspose3d(k3, s2) forward into a zero-filled window with a bias."""
    g = torch.Generator().manual_seed(6)
    t = bf16_values((1, cin, 4, 5, 6), g)
    if window:
        w = torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.05
        bias = C.dev(be, torch.randn(cout, generator=g))
        return _conv_both(be, t, w, 2, 3, (1, cout, 8, 10, 12), dict(stride=1, pad=1, in_mode=ops.IN_ZERO_INSERT, bias=bias, off=(0, 0, 1), out_dhw=(7, 9, 11)),
                          zero=True)
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.05          # the forward conv's weight [cin_fwd = cout here ...]: mode 1 pack
    res = acts(be, bf16_values((1, cout, 8, 9, 11), g))
    return _conv_both(be, t, w, 1, 3, (1, cout, 8, 9, 11), dict(stride=1, pad=1, in_mode=ops.IN_ZERO_INSERT, out_dhw=(8, 9, 11)), res=res)


def case_conv_k3_tile(be, cin, cout, dhw, norm=False, residual=False, drop=False, moments=False, gnb=False, mode=0, n=1):
    """3x3x3 stride-1 conv on the 16-bit-operand tile kernel (backend precision bf16) with 16-bit storage."""
    g = torch.Generator().manual_seed(7)
    t = bf16_values((n, cin, *dhw), g, 1.3, 0.2)
    w = torch.randn(*((cout, cin) if mode == 0 else (cin, cout)), 3, 3, 3, generator=g) * 0.05
    kw = {}
    if norm:
        x32, _ = acts(be, t)
        groups = 8 if cin % 8 == 0 else cin
        st = be.gn_stats(x32, groups, 1e-5, C.dev(be, torch.rand(cin, generator=g) + 0.5), C.dev(be, torch.randn(cin, generator=g) * 0.3))
        kw.update(in_mode=ops.IN_AFFINE_ACT, scale=st[1], shift=st[2])
    if drop:
        kw["chscale"] = C.dev(be, (torch.rand(n, cout, generator=g) > 0.3).float() / 0.7)
    if moments:
        kw["moments"] = True
    res = acts(be, bf16_values((n, cout, *dhw), g)) if residual else None
    gnb_of = None
    if gnb:
        gx = bf16_values((n, cout, *dhw), g, 1.5, 0.3)
        gx32, gx16 = acts(be, gx)
        groups = 8 if cout % 8 == 0 else cout
        st = be.gn_stats(gx16, groups, 1e-5, C.dev(be, torch.rand(cout, generator=g) + 0.5), C.dev(be, torch.randn(cout, generator=g) * 0.3))
        gnb_of = (gx32, gx16, st, groups)
    return _conv_both(be, t, w, mode, 3, (n, cout, *dhw), kw, res=res, gnb_of=gnb_of)


def case_first_layer(be, dhw=(6, 9, 10), cout=32):
    """4-channel fp32 input (the network input stays fp32): forward writes a bf16 tensor + its moment records; the weight gradient reads
    fp32 x with a bf16 upstream gradient; the narrow dgrad reads a bf16 gradient and writes the fp32 input gradient."""
    g = torch.Generator().manual_seed(8)
    n = 2
    t = torch.randn(n, 4, *dhw, generator=g) * 1.5 + 0.2                       # NOT bf16-representable: the input volume is fp32
    x32 = C.to_act(be, t)
    st = be.gn_stats(x32, 4, 1e-5, C.dev(be, torch.rand(4, generator=g) + 0.5), C.dev(be, torch.randn(4, generator=g) * 0.3))
    w = torch.randn(cout, 4, 3, 3, 3, generator=g) * 0.1
    out = _conv_both(be, t, w, 0, 3, (n, cout, *dhw), dict(in_mode=ops.IN_AFFINE_ACT, scale=st[1], shift=st[2], moments=True), x16_is_f32=True)
    dy32, dy16 = acts(be, bf16_values((n, cout, *dhw), g))
    dws = []
    for dy in (dy32, dy16):
        dw = torch.empty(cout, 4, 3, 3, 3, device=be.device)
        be.conv_wgrad(x32, dy, dw, 3, 1, in_mode=ops.IN_AFFINE_ACT, scale=st[1], shift=st[2])
        dws.append(dw.cpu())
    out["wgrad"] = C.rel_err(dws[1], dws[0])
    if cout == 32:
        # round 6: the fused backward (csrc/conv3d_c4_bwd.hip) reads the upstream gradient in its storage type: the same values as fp32
        # and as a 16-bit tensor must give the same dW / dgamma / dbeta (exact fp32 arithmetic on the stored values in both)
        gam = C.dev(be, torch.rand(4, generator=g) + 0.5)
        res = []
        for dy in (dy32, dy16):
            assert be.c4_bwd_supported(x32, dy, ops.IN_AFFINE_ACT, 0.0, st[1], st[2])
            dw = torch.full((cout, 4, 3, 3, 3), 7.0, device=be.device)
            dg, db = torch.full((4,), 7.0, device=be.device), torch.full((4,), 7.0, device=be.device)
            be.c4_bwd(x32, dy, be.pack_weight(C.dev(be, w), 1), dw, 4, gam, st[0], st[1], st[2], dg, db)
            res.append((dw.cpu(), dg.cpu(), db.cpu()))
        out["c4bwd_dw"], out["c4bwd_dgamma"], out["c4bwd_dbeta"] = (C.rel_err(a, b) for a, b in zip(res[1], res[0]))
        out["c4bwd_vs_wgrad"] = C.rel_err(res[0][0], dws[0])          # ... and its dW is the weight-gradient kernel's
    out.update({"dgrad_" + k: v for k, v in _conv_both(be, dy32.tensor().permute(0, 4, 1, 2, 3).cpu(), w, 1, 3, (n, 4, *dhw), {}, y16_is_f32=True).items()})
    return out


def case_wgrad(be, kd, stride, cin, cout, dhw, norm=False, n=1):
    g = torch.Generator().manual_seed(9)
    t = bf16_values((n, cin, *dhw), g, 1.3, 0.2)
    x32, x16 = acts(be, t, ld=cin + 8, c0=0)
    odhw = tuple((s - 1) // stride + 1 for s in dhw) if kd == 3 else dhw
    dy32, dy16 = acts(be, bf16_values((n, cout, *odhw), g))
    kw = {}
    if norm:
        groups = 8 if cin % 8 == 0 else cin
        st = be.gn_stats(x32, groups, 1e-5, C.dev(be, torch.rand(cin, generator=g) + 0.5), C.dev(be, torch.randn(cin, generator=g) * 0.3))
        kw = dict(in_mode=ops.IN_AFFINE_ACT, scale=st[1], shift=st[2])
    dws = []
    for x, dy in ((x32, dy32), (x16, dy16)):
        dw = torch.full((cout, cin, kd, kd, kd), 3.0, device=be.device)
        be.conv_wgrad(x, dy, dw, kd, stride, **kw)
        dws.append(dw.cpu())
    return {"dw": C.rel_err(dws[1], dws[0])}
