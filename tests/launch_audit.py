"""TEST INFRASTRUCTURE: per-launch audit of a whole training step (the conditioning-independent backward check).

Every kernel launch a network step makes through `ops.Backend` -- conv forward / dgrad (direct, Winograd, first-layer, zero-insert,
depth-to-space forms), weight gradients, norm statistics, norm+activation backward, trilinear up-sampling, Dropout3d scale, the
class projection, the Dice loss and Adam -- is recomputed in **fp64 on the CPU from the SAME input tensors the launch read** (torch
ATen ops following the reference's modules: resnet.py:12-22, myronenko.py:17-21,47-58,75-80, decoder.py:99-106, unet.py:27-44) and the
launch's output is compared with it. Errors therefore do not compound through the network and the ill-conditioning of whole-network
gradients does not enter: each launch is held to a bound that only its own arithmetic explains.

Cost control (the 128^3 headline step has ~600 launches): big tensors are audited on samples that keep the check exact --
  conv forward / dgrad : output BLOCKS (both corners of the volume, so every padding face is covered, + random interior blocks) with
                         all channels, from the haloed input crop of each block;
  weight gradients     : channel SUBSETS (ci, co) with the full voxel sum;
  norm statistics / norm backward : whole groups (all samples, all voxels) of a subset of groups;
  up-sampling, dropout scale      : channel subsets;   projection, Dice, Adam: complete.
The weight a conv launch is checked against is the PARAMETER tensor the pack was made from (PackedWeight.w), not the packed copy:
the pack kernels (direct and Winograd) are inside the audited path.

16-bit operand model (the mixed-precision modes "bf16" / "fp16" of BASELINE configs[2], reference: AutocastUNet, segmentation/unet.py:53-58):
the 3x3x3 stride-1 convolutions and their weight gradients round BOTH operands to the 16-bit type while staging them -- the activated
input act(scale * x + shift) evaluated in fp32, the weights, the upstream gradient -- and accumulate exact products in fp32. Such a
launch is recomputed in fp64 **from operands rounded the same way** (`tensor.bfloat16()` / `.half()`, round to nearest even), so the
comparison isolates the launch's own arithmetic (fp32 accumulation order, ~1e-6) from the mode's rounding (2^-9 / 2^-12 per operand,
which the op-level tolerance tests price): the bound stays 1e-5. The fp32 evaluation of the norm prologue is modelled in both of its
legal forms (fused multiply-add, as hipcc contracts it on the GPU; separate multiply and add, as the CPU emulator build has it): a
value that lands on the other side of a 16-bit rounding boundary moves an operand by one 16-bit ulp, which is exactly the kind of
difference this model must not hide behind a loose bound -- the smaller of the two errors is recorded together with the form that
produced it. The split modes (bf16x3 / bf16x6) keep fp32-class products and are held to their op-level bounds (1e-4 / 1e-5).
"""
import contextlib
import inspect
import math

import torch
import torch.nn.functional as F

from oracle import torch_ops as O

IN_PLAIN, IN_AFFINE_ACT, IN_ZERO_INSERT, IN_S2D = 0, 1, 2, 3
OUT_PLAIN, OUT_D2S = 0, 1
PREC_F32, PREC_BF16X3, PREC_BF16X6, PREC_BF16, PREC_F16 = 0, 1, 2, 3, 4        # include/mi355_unet3d.h MI355_PREC_*
LP_DTYPE = {PREC_BF16: torch.bfloat16, PREC_F16: torch.float16}


def round_lp(t64, dtype):
    """fp64 tensor holding fp32-representable values -> the same values rounded to `dtype` (RNE), back in fp64."""
    return t64.float().to(dtype).double()


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    den = float(b.abs().max())
    return float((a - b).abs().max()) / (den if den > 0 else 1.0)


def store_err(got, ref, dtype):
    """Error of a launch OUTPUT against its fp64 recomputation, relative to the block's largest value. fp32 tensors: rel_err. A tensor
    STORED as bf16 / fp16 (ops.Act.dtype, HipAutocastUNet(activation_storage=...)) holds the fp32 result rounded once, to nearest even: the
    rounding itself (up to half a bf16 ulp of each value) is the storage format, not an error of the launch -- what is left after it is
    taken out must meet the same bound as an fp32 output. A value the fp32 arithmetic puts on the other side of a rounding boundary
    differs by a whole ulp from round(ref): the half-ulp allowance around `ref` covers exactly the two candidates next to it."""
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    den = float(ref.abs().max())
    den = den if den > 0 else 1.0
    if dtype not in (torch.bfloat16, torch.float16):
        return float((got - ref).abs().max()) / den
    if dtype == torch.bfloat16:
        a = ref.abs().clamp(min=2.0 ** -126)
        ulp = torch.exp2(torch.floor(torch.log2(a)) - 7.0)       # spacing of bf16 (8 significand bits) at |ref|
    else:
        a = ref.abs().clamp(min=2.0 ** -14)                      # fp16: 11 significand bits, subnormal spacing 2^-24 below 2^-14
        ulp = torch.exp2(torch.floor(torch.log2(a)) - 10.0)
    return float(((got - ref).abs() - 0.5 * ulp).clamp(min=0.0).max()) / den


def _nc(t5):
    """[N, D, H, W, C] (any device) -> NCDHW fp64 on the CPU."""
    return t5.detach().cpu().double().permute(0, 4, 1, 2, 3).contiguous()


def _subset(c, k, rng):
    """<= k distinct channel indices of range(c): first, last and random ones, sorted."""
    if c <= k:
        return list(range(c))
    idx = {0, c - 1}
    while len(idx) < k:
        idx.add(int(torch.randint(0, c, (1,), generator=rng)))
    return sorted(idx)


class LaunchAudit:
    """with LaunchAudit(be) as au: <run a step through modules bound to `be`>;  au.records = [dict(kind, desc, err), ...]"""

    WRAPPED = ("conv_fwd", "conv_wgrad", "c4_bwd", "gn_stats", "gn_act_bwd", "upsample2x_fwd", "upsample2x_bwd", "chscale", "proj_fwd", "proj_bwd",
               "dice", "adam_step", "ncdhw_to_ndhwc", "ndhwc_to_ncdhw", "add", "cast")

    def __init__(self, be, block_macs=1.5e8, n_blocks=4, wgrad_channels=4, full_macs=4e8, seed=0, verbose=False):
        self.be = be
        self.block_macs, self.n_blocks, self.wch, self.full_macs = block_macs, n_blocks, wgrad_channels, full_macs
        self.rng = torch.Generator().manual_seed(seed)
        self.records = []
        self.verbose = verbose
        self._orig = {}

    # ---- plumbing ----------------------------------------------------------------------------------------------------------------
    def __enter__(self):
        for name in self.WRAPPED:
            orig = getattr(self.be, name)
            self._orig[name] = (orig, name in vars(self.be))
            setattr(self.be, name, self._wrap(name, orig))
        return self

    def __exit__(self, *exc):
        for name in self.WRAPPED:
            orig, was_instance_attr = self._orig[name]
            if was_instance_attr:
                setattr(self.be, name, orig)  # a test's own wrapper around the method
            else:
                delattr(self.be, name)        # the instance attribute shadowed the class method
        return False

    def _wrap(self, name, orig):
        sig = inspect.signature(orig)
        handler = getattr(self, "_a_" + name)

        def wrapped(*args, **kw):
            ba = sig.bind(*args, **kw)
            ba.apply_defaults()
            p = dict(ba.arguments)
            extra = p.pop("kw", None)            # methods that take **kw
            if extra:
                p.update(extra)
            return handler(orig, args, kw, p)
        return wrapped

    def _rec(self, kind, desc, err):
        self.records.append(dict(kind=kind, desc=desc, err=float(err)))
        if self.verbose:
            print(f"#{len(self.records):4d} {kind:14s} {err:.2e}  {desc}", flush=True)

    def worst(self):
        out = {}
        for r in self.records:
            if r["kind"] not in out or r["err"] > out[r["kind"]]["err"]:
                out[r["kind"]] = r
        return out

    def counts(self):
        out = {}
        for r in self.records:
            out[r["kind"]] = out.get(r["kind"], 0) + 1
        return out

    # ---- helpers -----------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _prologue(t, n_sel, c_sel, in_mode, slope, scale, shift, in_slope, f32_form=None):
        """act(scale * x + shift) of the conv / wgrad prologue on an NCDHW fp64 crop holding samples n_sel, channels c_sel.
        f32_form None: evaluated in fp64 (the fp32-precision launches: their fp32 evaluation is within the 1e-5 bound of this).
        "fma" / "muladd": evaluated AS THE KERNEL DOES in fp32 -- fl(x * s + h) in one rounding, or fl(fl(x * s) + h) -- then
        max(u, fl(u * slope)); the result (fp32 values held in fp64) is what a 16-bit launch rounds to its operand type."""
        if in_mode != IN_AFFINE_ACT or scale is None:
            return t
        sc = scale.detach().cpu().double()[n_sel][:, c_sel][:, :, None, None, None]
        sh = shift.detach().cpu().double()[n_sel][:, c_sel][:, :, None, None, None]
        if in_slope is not None:
            sl = in_slope.detach().cpu().double()[c_sel][None, :, None, None, None]
        else:
            sl = torch.tensor(float(torch.tensor(float(slope), dtype=torch.float32)), dtype=torch.float64)
        if f32_form is None:
            u = t * sc + sh
            return torch.maximum(u, u * sl)      # the kernels' form of (leaky) ReLU for slope in [0, 1]
        # fp32 operands: the product of two fp32 values is exact in fp64 (48 significand bits), so .float() of the fp64 expression is
        # the correctly rounded fp32 result of each form (double rounding only on exact fp64 ties of the sum: never seen)
        if f32_form == "fma":
            u = (t * sc + sh).float().double()
        else:
            u = ((t * sc).float().double() + sh).float().double()
        return torch.maximum(u, (u * sl).float().double())

    @staticmethod
    def _weight(wp):
        """Correlation weight [out, in, k, k, k] (fp64, CPU) of a PackedWeight, from the tensor it was packed from (conv3d_fwd.hip pack modes)."""
        w = wp.w.detach().cpu().double()
        if wp.mode in (1, 2):
            w = w.transpose(0, 1).flip(2, 3, 4)
        return w.contiguous()

    def _eff_input_block(self, x, p, lo, hi, c_sel=None, f32_form=None):
        """Effective (activated, zero-padded, zero-inserted) conv input over the effective coordinates [lo, hi) per axis, NCDHW fp64."""
        n = x.shape[0]
        c_sel = list(range(x.c)) if c_sel is None else c_sel
        zi = p["in_mode"] == IN_ZERO_INSERT
        dims = x.shape[1:4]
        src, dst = [], []
        for ax in range(3):
            e = torch.arange(lo[ax], hi[ax])
            if zi:
                ok = (e % 2 == 0) & (e >= 0) & (e // 2 < dims[ax])
                s = e[ok] // 2
            else:
                ok = (e >= 0) & (e < dims[ax])
                s = e[ok]
            src.append(s)
            dst.append(torch.nonzero(ok).flatten())
        E = torch.zeros(n, len(c_sel), *[hi[a] - lo[a] for a in range(3)], dtype=torch.float64)
        if all(len(s) for s in src):
            b0 = [int(s.min()) for s in src]
            b1 = [int(s.max()) + 1 for s in src]
            crop = x.tensor()[:, b0[0]:b1[0], b0[1]:b1[1], b0[2]:b1[2], :]
            if len(c_sel) != x.c:
                crop = crop[..., torch.as_tensor(c_sel, device=crop.device)]
            crop = _nc(crop)
            crop = self._prologue(crop, slice(None), c_sel, p["in_mode"], p.get("slope", 0.0), p.get("scale"), p.get("shift"), p.get("in_slope"),
                                  f32_form)
            sub = crop[:, :, (src[0] - b0[0])[:, None, None], (src[1] - b0[1])[None, :, None], (src[2] - b0[2])[None, None, :]]
            E[:, :, dst[0][:, None, None], dst[1][None, :, None], dst[2][None, None, :]] = sub
        return E

    # ---- arithmetic of a launch ------------------------------------------------------------------------------------------------------
    def _fwd_precision(self, p, x, y, kd, stride, pad):
        """Backend precision mode a conv_fwd launch computes in (routing of ops.PackedWeight.ptr_for / mi355_conv3d_uses_bf16):
        the opt-in modes apply to the 3x3x3 stride-1 forward / dgrad convolutions with a plain or norm-prologue input, except the
        first layer's dgrad (<= 4 output channels: the exact-fp32 vector-ALU kernel in every mode)."""
        prec = int(getattr(self.be, "precision", PREC_F32))
        if prec == PREC_F32 or kd != 3 or stride != 1 or p["in_mode"] not in (IN_PLAIN, IN_AFFINE_ACT):
            return PREC_F32
        first_layer = p["wp"].mode == 0 and x.c == 4 and pad == 1 and p["out_mode"] == OUT_PLAIN
        narrow = (y.c <= 4 and pad == 1 and p["in_mode"] == IN_PLAIN and p["out_mode"] == OUT_PLAIN and p["bias"] is None
                  and p["residual"] is None and p["chscale"] is None and tuple(p["off"]) == (0, 0, 0))
        return PREC_F32 if (narrow and not first_layer) else prec

    def _wgrad_precision(self, p, x, kd, stride, pad):
        """... a conv_wgrad launch computes in (mi355_conv3d_wgrad: the 4-input-channel first layer stays exact fp32)."""
        prec = int(getattr(self.be, "precision", PREC_F32))
        if (prec == PREC_F32 or kd != 3 or stride != 1 or pad != 1 or p["out_mode"] != OUT_PLAIN or p["in_mode"] not in (IN_PLAIN, IN_AFFINE_ACT)
                or x.c == 4):
            return PREC_F32
        return prec

    def _blocks(self, lo, hi, edge):
        """<= n_blocks blocks [(z0,z1),(y0,y1),(x0,x1)] inside [lo, hi): the low corner, the high corner, random ones (one block when it all fits)."""
        ext = [hi[a] - lo[a] for a in range(3)]
        if all(e <= edge for e in ext):
            return [[(lo[a], hi[a]) for a in range(3)]]
        out = []
        for k in range(self.n_blocks):
            blk = []
            for a in range(3):
                b = min(edge, ext[a])
                if k == 0:
                    s = lo[a]
                elif k == 1:
                    s = hi[a] - b
                else:
                    s = lo[a] + int(torch.randint(0, ext[a] - b + 1, (1,), generator=self.rng))
                blk.append((s, s + b))
            out.append(blk)
        return out

    # ---- conv forward / dgrad ----------------------------------------------------------------------------------------------------
    def _a_conv_fwd(self, orig, args, kw, p):
        x, wp, y, kd = p["x"], p["wp"], p["y"], p["kd"]
        in_mode, out_mode = p["in_mode"], p["out_mode"]
        res_snapshot = None
        ret = orig(*args, **kw)
        stride = p["stride"]
        pad = kd // 2 if p["pad"] is None else p["pad"]
        W = self._weight(wp)
        n = x.shape[0]
        desc = f"k{kd} s{stride} in{in_mode} out{out_mode} {x.c}->{y.c} @{tuple(y.shape[1:4])} {'res ' if p['residual'] is not None else ''}" \
               f"{'drop ' if p['chscale'] is not None else ''}{'bias ' if p['bias'] is not None else ''}{type(wp).__name__}"
        if out_mode == OUT_D2S or in_mode == IN_S2D:
            self._rec("conv_d2s", desc, self._conv_d2s(x, W, y, p))
            return ret
        off = tuple(p["off"])
        out_dhw = tuple(p["out_dhw"]) if p["out_dhw"] is not None else tuple(y.shape[1:4])
        ydims = y.shape[1:4]
        lo = [max(0, -off[a]) for a in range(3)]
        hi = [min(out_dhw[a], ydims[a] - off[a]) for a in range(3)]
        s_eff = 1 if in_mode == IN_ZERO_INSERT else stride
        macs_per_vox = float(x.c) * y.c * kd ** 3 / (8.0 if in_mode == IN_ZERO_INSERT else 1.0)
        edge = int(max(3, min(14, round((self.block_macs / (n * macs_per_vox)) ** (1.0 / 3.0)))))
        if n * macs_per_vox * math.prod(hi[a] - lo[a] for a in range(3)) <= self.full_macs:
            edge = 1 << 30
        prec = self._fwd_precision(p, x, y, kd, stride, pad)
        lp = LP_DTYPE.get(prec)
        Wc = W[:y.c, :x.c]
        kind = "conv_fwd"
        forms = [None]
        if lp is not None:
            # 16-bit operand model: both operands rounded to the operand type, exact products, fp64 sum (module docstring)
            Wc = round_lp(Wc, lp)
            kind = "conv_fwd_lp"
            forms = ["fma", "muladd"] if (in_mode == IN_AFFINE_ACT and p.get("scale") is not None) else ["fma"]
        elif prec != PREC_F32:
            kind = {PREC_BF16X3: "conv_fwd_x3", PREC_BF16X6: "conv_fwd_x6"}[prec]
        worst = {f: 0.0 for f in forms}
        for blk in self._blocks(lo, hi, edge):
            elo = [blk[a][0] * s_eff - pad for a in range(3)]
            ehi = [(blk[a][1] - 1) * s_eff - pad + kd for a in range(3)]
            ysl = tuple(slice(off[a] + blk[a][0], off[a] + blk[a][1]) for a in range(3))
            got = _nc(y.tensor()[(slice(None),) + ysl])
            for form in forms:
                E = self._eff_input_block(x, p, elo, ehi, f32_form=form)
                if lp is not None:
                    E = round_lp(E, lp)
                ref = F.conv3d(E, Wc, None, stride=s_eff)
                if p["bias"] is not None:
                    ref = ref + p["bias"].detach().cpu().double()[None, :, None, None, None]
                if p["residual"] is not None:
                    ref = ref + _nc(p["residual"].tensor()[(slice(None),) + ysl])
                if p["chscale"] is not None:
                    ref = ref * p["chscale"].detach().cpu().double()[:, :, None, None, None]
                # error of the block relative to the magnitude of the block itself (a stricter denominator than the tensor's maximum)
                worst[form] = max(worst[form], store_err(got, ref, y.dtype))
        form = min(worst, key=worst.get)
        self._rec(kind, desc + (f" [{lp} operands, prologue {form}]" if lp is not None else "") + (f" [{'bf16' if y.dtype == torch.bfloat16 else 'fp16'} storage]" if y.dtype in (torch.bfloat16, torch.float16) else ""),
                  worst[form])
        return ret

    def _conv_d2s(self, x, W, y, p):
        """ConvTranspose3d(k2, s2) as a 1x1x1 GEMM: depth-to-space epilogue (forward) / space-to-depth prologue (its dgrad). Complete
        tensors when small, else a z-slab of coarse planes."""
        n = x.shape[0]
        if p["out_mode"] == OUT_D2S:
            D = x.shape[1]
            z0, z1 = (0, D) if x.shape[1] * x.shape[2] * x.shape[3] * x.c * 8 * y.c <= self.full_macs else (D // 2, D // 2 + 2)
            t = _nc(x.tensor()[:, z0:z1])
            t = self._prologue(t, slice(None), list(range(x.c)), p["in_mode"], p.get("slope", 0.0), p.get("scale"), p.get("shift"), p.get("in_slope"))
            g = F.conv3d(t, W)                                     # [n, 8*co, d, h, w], rows (p = 4a+2b+e, co)
            co = y.c
            d, h, w = g.shape[2:]
            g = g.view(n, 2, 2, 2, co, d, h, w).permute(0, 4, 5, 1, 6, 2, 7, 3).reshape(n, co, 2 * d, 2 * h, 2 * w)
            if p["bias"] is not None:
                g = g + p["bias"].detach().cpu().double()[None, :, None, None, None]
            return rel_err(_nc(y.tensor()[:, 2 * z0:2 * z1]), g)
        # IN_S2D: x is the fine gradient [n, 2D, 2H, 2W, co], y the coarse [n, D, H, W, ci]
        D = y.shape[1]
        z0, z1 = (0, D) if y.shape[1] * y.shape[2] * y.shape[3] * y.c * 8 * x.c <= self.full_macs else (D // 2, D // 2 + 2)
        f = _nc(x.tensor()[:, 2 * z0:2 * z1])
        co = x.c
        d, h, w = f.shape[2] // 2, f.shape[3] // 2, f.shape[4] // 2
        s = f.view(n, co, d, 2, h, 2, w, 2).permute(0, 3, 5, 7, 1, 2, 4, 6).reshape(n, 8 * co, d, h, w)
        ref = F.conv3d(s, W)
        if p["residual"] is not None:
            ref = ref + _nc(p["residual"].tensor()[:, z0:z1])
        return rel_err(_nc(y.tensor()[:, z0:z1]), ref)

    # ---- weight gradient ---------------------------------------------------------------------------------------------------------
    def _a_conv_wgrad(self, orig, args, kw, p):
        x, dy, dw, kd = p["x"], p["dy"], p["dw"], p["kd"]
        ret = orig(*args, **kw)
        stride = p["stride"]
        pad = kd // 2 if p["pad"] is None else p["pad"]
        n = x.shape[0]
        desc = f"k{kd} s{stride} in{p['in_mode']} out{p['out_mode']} {x.c}->{dy.c} @{tuple(dy.shape[1:4])}"
        if p["out_mode"] == OUT_D2S:
            # dw [(p, co), ci]: the GEMM rows are the 8 fine-voxel parities
            ci = _subset(x.c, self.wch, self.rng)
            co = _subset(dy.c, self.wch, self.rng)
            t = _nc(x.tensor()[..., torch.as_tensor(ci, device=x.buf.device)])
            t = self._prologue(t, slice(None), ci, p["in_mode"], p.get("slope", 0.0), p.get("scale"), p.get("shift"), p.get("in_slope"))
            f = _nc(dy.tensor()[..., torch.as_tensor(co, device=dy.buf.device)])
            d, h, w = t.shape[2:]
            s = f.view(n, len(co), d, 2, h, 2, w, 2).permute(3, 5, 7, 1, 0, 2, 4, 6).reshape(8, len(co), -1)
            ref = torch.einsum("pkv,cv->pkc", s, t.permute(1, 0, 2, 3, 4).reshape(len(ci), -1))
            got = dw.detach().cpu().double().view(8, dy.c, x.c)[:, co][:, :, ci]
            self._rec("conv_wgrad", desc, rel_err(got, ref) if float(ref.abs().max()) > 0 else 0.0)
            return ret
        # the reference holds the full voxel sum of every audited (ci, co) pair in fp64: 27 strided copies of the input subset. On the
        # largest tensors (128^3 x 4 of BASELINE configs[2]: 8.4 M voxels) two channels per side keep a launch's audit near one second
        wch = self.wch if n * dy.shape[1] * dy.shape[2] * dy.shape[3] <= 6e6 else min(self.wch, 2)
        ci = _subset(x.c, wch, self.rng)
        co = _subset(dy.c, wch, self.rng)
        prec = self._wgrad_precision(p, x, kd, stride, pad)
        lp = LP_DTYPE.get(prec)
        kind = "conv_wgrad"
        forms = [None]
        if lp is not None:
            kind = "conv_wgrad_lp"
            forms = ["fma", "muladd"] if (p["in_mode"] == IN_AFFINE_ACT and p.get("scale") is not None) else ["fma"]
        elif prec != PREC_F32:
            kind = {PREC_BF16X3: "conv_wgrad_x3", PREC_BF16X6: "conv_wgrad_x6"}[prec]
        t0 = _nc(x.tensor()[..., torch.as_tensor(ci, device=x.buf.device)])
        g = _nc(dy.tensor()[..., torch.as_tensor(co, device=dy.buf.device)])
        if lp is not None:
            g = round_lp(g, lp)                                # the upstream gradient is the second operand of the weight-gradient product
        Do, Ho, Wo = g.shape[2:]
        need = [(Do - 1) * stride + kd, (Ho - 1) * stride + kd, (Wo - 1) * stride + kd]
        gf = g.permute(1, 0, 2, 3, 4).reshape(len(co), -1)
        got = dw.detach().cpu().double().reshape(dy.c, -1, kd, kd, kd)[co][:, ci]
        # the tensor-level scale: max |dw| over the whole gradient (the subset's own maximum can sit far below it)
        scale = max(float(dw.detach().abs().max()), 1e-300)
        errs = {}
        for form in forms:
            t = self._prologue(t0, slice(None), ci, p["in_mode"], p.get("slope", 0.0), p.get("scale"), p.get("shift"), p.get("in_slope"), form)
            if lp is not None:
                t = round_lp(t, lp)
            tp = F.pad(t, [pad, max(0, need[2] - pad - t.shape[4]), pad, max(0, need[1] - pad - t.shape[3]), pad, max(0, need[0] - pad - t.shape[2])])
            ref = torch.empty(len(co), len(ci), kd, kd, kd, dtype=torch.float64)
            for a in range(kd):
                for b in range(kd):
                    for c in range(kd):
                        xs = tp[:, :, a:a + (Do - 1) * stride + 1:stride, b:b + (Ho - 1) * stride + 1:stride, c:c + (Wo - 1) * stride + 1:stride]
                        ref[:, :, a, b, c] = gf @ xs.permute(1, 0, 2, 3, 4).reshape(len(ci), -1).t()
            errs[form] = float((got - ref).abs().max()) / scale
        form = min(errs, key=errs.get)
        self._rec(kind, desc + (f" [{lp} operands, prologue {form}]" if lp is not None else ""), errs[form])
        return ret

    def _a_c4_bwd(self, orig, args, kw, p):
        """The fused first-layer backward (csrc/conv3d_c4_bwd.hip): dW of the 4 -> 32 conv (complete) and dgamma / dbeta of the norm in front
        of it, from the launch's own inputs in fp64. Recorded as one conv_wgrad and one gn_act_bwd launch (the kinds it replaces)."""
        x, dy, dw, groups = p["x"], p["dy"], p["dw"], p["groups"]
        ret = orig(*args, **kw)
        t0, g = _nc(x.tensor()), _nc(dy.tensor())
        a = self._prologue(t0, slice(None), list(range(4)), IN_AFFINE_ACT, p["slope"], p["scale"], p["shift"], p["in_slope"])
        w = self._weight(p["wp"]).transpose(0, 1).flip(2, 3, 4).contiguous()      # back to the forward weight [32, 4, 3, 3, 3]
        dw_ref = torch.nn.grad.conv3d_weight(a, (32, 4, 3, 3, 3), g, padding=1)
        e_w = float((dw.detach().cpu().double() - dw_ref).abs().max()) / max(float(dw_ref.abs().max()), 1e-300)
        self._rec("conv_wgrad", f"c4_bwd k3 s1 4->32 @{tuple(dy.shape[1:4])}", e_w)
        dA = torch.nn.grad.conv3d_input(a.shape, w, g, padding=1)
        sc = p["scale"].detach().cpu().double()[:, :, None, None, None]
        sh = p["shift"].detach().cpu().double()[:, :, None, None, None]
        u = t0 * sc + sh                                     # the mask the kernel uses: its own scale / shift
        du = dA * torch.where(u > 0, torch.ones_like(u), torch.full_like(u, float(p["slope"])))
        mr = p["mean_rstd"].detach().cpu().double()
        cpg = 4 // groups
        mean = mr[..., 0].repeat_interleave(cpg, 1)[:, :, None, None, None]
        rstd = mr[..., 1].repeat_interleave(cpg, 1)[:, :, None, None, None]
        dgr, dbr = (du * (t0 - mean) * rstd).sum((0, 2, 3, 4)), du.sum((0, 2, 3, 4))
        dgk, dbk = p["dgamma"].detach().cpu().double(), p["dbeta"].detach().cpu().double()
        e_g = float((dgk - dgr).abs().max()) / max(float(dgr.abs().max()), 1e-300)
        e_b = float((dbk - dbr).abs().max()) / max(float(dbr.abs().max()), 1e-300)
        self._rec("gn_act_bwd", f"c4_bwd C=4 G={groups} @{tuple(x.shape[1:4])} fused=True", max(e_g, e_b))
        return ret

    # ---- norm --------------------------------------------------------------------------------------------------------------------
    def _group_subset(self, c, groups, budget=8):
        cpg = c // groups
        gs = _subset(groups, max(2, budget // cpg), self.rng)
        ch = [g * cpg + k for g in gs for k in range(cpg)]
        return gs, ch

    def _a_gn_stats(self, orig, args, kw, p):
        x, groups, eps = p["x"], p["groups"], p["eps"]
        mr, sc, sh = orig(*args, **kw)
        gs, ch = self._group_subset(x.c, groups)
        t = _nc(x.tensor()[..., torch.as_tensor(ch, device=x.buf.device)])
        n = t.shape[0]
        g = t.reshape(n, len(gs), -1)
        mean, var = g.mean(-1), g.var(-1, unbiased=False)
        rstd = (var + eps).rsqrt()
        cpg = x.c // groups
        ga = p["gamma"].detach().cpu().double()[ch] if p["gamma"] is not None else torch.ones(len(ch), dtype=torch.float64)
        be_ = p["beta"].detach().cpu().double()[ch] if p["beta"] is not None else torch.zeros(len(ch), dtype=torch.float64)
        scr = ga[None] * rstd.repeat_interleave(cpg, 1)
        shr = be_[None] - mean.repeat_interleave(cpg, 1) * scr
        mrc = mr.detach().cpu().double()[:, gs]
        # mean is held to the group's standard deviation (a mean of 1e-7 sigma is zero to fp32), rstd / scale / shift relatively
        e_mean = float(((mrc[..., 0] - mean).abs() * rstd).max())
        e = max(e_mean, rel_err(mrc[..., 1], rstd), rel_err(sc.detach().cpu().double()[:, ch], scr),
                float((sh.detach().cpu().double()[:, ch] - shr).abs().max()) / max(float(shr.abs().max()), float(scr.abs().max())))
        self._rec("gn_stats", f"C={x.c} G={groups} @{tuple(x.shape[1:4])} fused={x.mom is not None}", e)
        return mr, sc, sh

    @torch.enable_grad()
    def _a_gn_act_bwd(self, orig, args, kw, p):
        x, dA, dx, groups, slope = p["x"], p["dA"], p["dx"], p["groups"], p["slope"]
        gs, ch = self._group_subset(x.c, groups)
        cpg = x.c // groups
        sel = torch.as_tensor(ch, device=x.buf.device)
        xin = _nc(x.tensor()[..., sel]).requires_grad_(True)
        dAi = _nc(dA.tensor()[..., sel])                     # before the launch: dx may alias dA
        add = _nc(p["addend"].tensor()[..., sel]) if p["addend"] is not None else None
        ret = orig(*args, **kw)
        sc = p["scale"].detach().cpu().double()[:, ch][:, :, None, None, None]
        sh = p["shift"].detach().cpu().double()[:, ch][:, :, None, None, None]
        u = xin.detach() * sc + sh                           # the mask the kernel uses: its own scale / shift
        mask = torch.where(u > 0, torch.ones_like(u), torch.full_like(u, float(slope)))
        g = p["gamma"].detach().cpu().double()[ch].requires_grad_(True)
        b = torch.zeros_like(g).requires_grad_(True)
        out = F.group_norm(xin, len(gs), g, b, 1e-5)
        dxr, dgr, dbr = torch.autograd.grad(out, (xin, g, b), dAi * mask)
        if add is not None:
            dxr = dxr + add
        e_dx = store_err(_nc(dx.tensor()[..., sel]), dxr, dx.dtype)
        # gamma / beta gradients relative to the largest gradient of the whole vector (a channel's own value can cancel to ~0)
        dgk, dbk = p["dgamma"].detach().cpu().double(), p["dbeta"].detach().cpu().double()
        e_g = float((dgk[ch] - dgr).abs().max()) / max(float(dgk.abs().max()), 1e-300)
        e_b = float((dbk[ch] - dbr).abs().max()) / max(float(dbk.abs().max()), 1e-300)
        self._rec("gn_act_bwd", f"C={x.c} G={groups} @{tuple(x.shape[1:4])} fused={p['partials'] is not None} cpg={cpg}", max(e_dx, e_g, e_b))
        return ret

    # ---- pointwise ---------------------------------------------------------------------------------------------------------------
    def _a_upsample2x_fwd(self, orig, args, kw, p):
        lo, cat, off = p["lo"], p["cat"], p["off"]
        ret = orig(*args, **kw)
        ch = _subset(lo.c, 4, self.rng)
        sel = torch.as_tensor(ch, device=lo.buf.device)
        ref = O.upsample_pad(_nc(lo.tensor()[..., sel]), cat.shape[1:4])
        self._rec("upsample_fwd", f"C={lo.c} @{tuple(cat.shape[1:4])}", store_err(_nc(cat.tensor()[..., sel]), ref, cat.dtype))
        return ret

    @torch.enable_grad()
    def _a_upsample2x_bwd(self, orig, args, kw, p):
        dcat, dlo = p["dcat"], p["dlo"]
        ret = orig(*args, **kw)
        ch = _subset(dlo.c, 4, self.rng)
        sel = torch.as_tensor(ch, device=dlo.buf.device)
        lo = torch.zeros(dlo.shape[0], len(ch), *dlo.shape[1:4], dtype=torch.float64, requires_grad=True)
        (ref,) = torch.autograd.grad(O.upsample_pad(lo, dcat.shape[1:4]), lo, _nc(dcat.tensor()[..., sel]))
        self._rec("upsample_bwd", f"C={dlo.c} @{tuple(dcat.shape[1:4])}", store_err(_nc(dlo.tensor()[..., sel]), ref, dlo.dtype))
        return ret

    def _a_chscale(self, orig, args, kw, p):
        x, s, y = p["x"], p["s"], p["y"]
        ch = _subset(x.c, 4, self.rng)
        sel = torch.as_tensor(ch, device=x.buf.device)
        before = _nc(x.tensor()[..., sel])                   # in place in the Dropout3d backward
        ret = orig(*args, **kw)
        ref = before * s.detach().cpu().double()[:, ch][:, :, None, None, None]
        self._rec("chscale", f"C={x.c} @{tuple(x.shape[1:4])}", store_err(_nc(y.tensor()[..., sel]), ref, y.dtype))
        return ret

    def _a_add(self, orig, args, kw, p):
        a, b, y = p["a"], p["b"], p["y"]
        ch = _subset(a.c, 4, self.rng)
        sel = torch.as_tensor(ch, device=a.buf.device)
        ref = _nc(a.tensor()[..., sel]) + _nc(b.tensor()[..., sel])
        ret = orig(*args, **kw)
        self._rec("add", f"C={a.c}", store_err(_nc(y.tensor()[..., sel]), ref, y.dtype))
        return ret

    def _a_cast(self, orig, args, kw, p):
        ret = orig(*args, **kw)
        ok = torch.equal(ret.tensor(), p["x"].tensor().to(ret.dtype))        # bit-exact: torch rounds to nearest even as v_cvt_pk_bf16_f32 does
        self._rec("layout", f"cast {p['x'].dtype} -> {ret.dtype}", 0.0 if ok else 1.0)
        return ret

    def _a_ncdhw_to_ndhwc(self, orig, args, kw, p):
        ret = orig(*args, **kw)
        ok = torch.equal(p["dst"].tensor().permute(0, 4, 1, 2, 3), p["src"].to(p["dst"].dtype))      # (a 16-bit destination: rounded once)
        self._rec("layout", "ncdhw->ndhwc", 0.0 if ok else 1.0)
        return ret

    def _a_ndhwc_to_ncdhw(self, orig, args, kw, p):
        ret = orig(*args, **kw)
        ok = torch.equal(p["src"].tensor().permute(0, 4, 1, 2, 3).float(), p["dst"])
        self._rec("layout", "ndhwc->ncdhw", 0.0 if ok else 1.0)
        return ret

    # ---- head, loss, optimizer ---------------------------------------------------------------------------------------------------
    def _act_full(self, x, scale, shift, slope):
        t = _nc(x.tensor())
        if scale is None:
            return t
        u = t * scale.detach().cpu().double()[:, :, None, None, None] + shift.detach().cpu().double()[:, :, None, None, None]
        return torch.maximum(u, u * float(slope))

    def _a_proj_fwd(self, orig, args, kw, p):
        ret = orig(*args, **kw)
        t = self._act_full(p["x"], p["scale"], p["shift"], p["slope"])
        ref = torch.einsum("ncdhw,kc->nkdhw", t, p["w"].detach().cpu().double())
        if p["bias"] is not None:
            ref = ref + p["bias"].detach().cpu().double()[None, :, None, None, None]
        self._rec("proj_fwd", f"{p['x'].c}->{p['w'].shape[0]} @{tuple(p['x'].shape[1:4])}", rel_err(p["logits"], ref))
        return ret

    def _a_proj_bwd(self, orig, args, kw, p):
        ret = orig(*args, **kw)
        t = self._act_full(p["x"], p["scale"], p["shift"], p["slope"])
        dz = p["dlogits"].detach().cpu().double()
        e = [rel_err(p["dw"], torch.einsum("nkdhw,ncdhw->kc", dz, t))]
        if p["dx"] is not None and p["scale"] is None:
            e.append(store_err(_nc(p["dx"].tensor()), torch.einsum("nkdhw,kc->ncdhw", dz, p["w"].detach().cpu().double()), p["dx"].dtype))
        if p["dbias"] is not None:
            e.append(rel_err(p["dbias"], dz.sum(dim=(0, 2, 3, 4))))
        self._rec("proj_bwd", f"{p['x'].c}->{p['w'].shape[0]} @{tuple(p['x'].shape[1:4])}", max(e))
        return ret

    @torch.enable_grad()
    def _a_dice(self, orig, args, kw, p):
        loss, d = orig(*args, **kw)
        if p.get("generalized"):
            return loss, d
        z = p["logits"].detach().cpu().double().requires_grad_(True)
        l = O.dice_loss(z, p["target"].cpu(), p["sigmoid"], p["batch"], p["squared_pred"], p["smooth_nr"], p["smooth_dr"], p["include_background"])
        l.backward()
        e = abs(float(loss.detach()) - float(l.detach())) / abs(float(l.detach()))
        if d is not None:
            e = max(e, rel_err(d, z.grad * float(p["grad_scale"])))
        self._rec("dice", f"{tuple(p['logits'].shape)}", e)
        return loss, d

    def _a_adam_step(self, orig, args, kw, p):
        pr, g, m, v = (p[k].detach().cpu().double() for k in ("p", "g", "m", "v"))
        ret = orig(*args, **kw)
        lr, b1, b2, eps, wd, step, gs = p["lr"], p["beta1"], p["beta2"], p["eps"], p["weight_decay"], p["step"], p["grad_scale"]
        gg = g * gs + wd * pr                                  # torch.optim.Adam (L2 form), bias-corrected
        m2 = b1 * m + (1 - b1) * gg
        v2 = b2 * v + (1 - b2) * gg * gg
        upd = (lr / (1 - b1 ** step)) * m2 / (v2.sqrt() / math.sqrt(1 - b2 ** step) + eps)
        pn = p["p"].detach().cpu().double()
        self._rec("adam", f"{pr.numel()} params step {step}", max(rel_err(pn, pr - upd), rel_err(p["m"], m2), rel_err(p["v"], v2)))
        # the update itself, relative to lr (p - update agrees to the last bit of p almost by construction): bounded below by the fp32
        # spacing of p, ~6e-8 * max|p| / lr
        self._rec("adam_update", f"{pr.numel()} params step {step}", float(((pr - pn) - upd).abs().max()) / lr)
        return ret


@contextlib.contextmanager
def audited(be, **kw):
    au = LaunchAudit(be, **kw)
    with au:
        yield au
