"""HipDynUNet: drop-in for MONAI `DynUNet` as the reference's shipped configs select it
(examples/brats2020/brats2020_config.json:2-107 -> unet3d/models/build.py:9-13 via `from monai.networks.nets import *`,
unet3d/models/pytorch/__init__.py:1). MONAI is an un-vendored, unpinned dependency of the reference (requirements.txt:4)
and is not installed here: the structure below follows SURVEY.md section 8a-B / appendix D (MONAI >= 1.2 `DynUNet`,
`UnetBasicBlock`, `UnetUpBlock`, `UnetOutBlock`), parity is against the plain-torch restatement oracle/dynunet_ref.py
("parity unpinned": no MONAI golden vectors exist in the reference).

  UnetBasicBlock(ci, co, s): Conv3d(k3, stride s, pad 1, no bias) -> InstanceNorm3d(affine) -> LeakyReLU(0.01), twice
  UnetUpBlock(ci, co):       ConvTranspose3d(k2, s2, no bias) -> cat((up, skip), 1) -> UnetBasicBlock(2co, co, 1)
  UnetOutBlock:              Conv3d(filters[0] -> out_channels, k1, bias)

MI355X execution: every conv writes its RAW output; InstanceNorm statistics are one streaming pass (mi355_gn_stats with
G == C) and the normalise + affine + LeakyReLU is applied by the CONSUMER while it stages its LDS tile (conv prologue), so
normalised tensors never exist in HBM. The skip-concat is a channel slice of one buffer: [0, co) = raw transposed-conv output
(identity prologue: scale 1, shift 0, slope 1), [co, 2co) = the encoder block's raw output with its own norm prologue.
ConvTranspose3d(k2, s2) is ONE 1x1x1 GEMM with 8*co logical output channels whose epilogue scatters depth-to-space
(MI355_OUT_D2S); its dgrad / wgrad read the fine gradient through the space-to-depth view (MI355_IN_S2D / OUT_D2S).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ._lib import IN_AFFINE_ACT, IN_PLAIN, IN_S2D, IN_ZERO_INSERT, OUT_D2S
from .engine import HipNetBase

IN_EPS = 1e-5
SLOPE = 0.01


def _all_equal(v, val):
    if isinstance(v, (list, tuple)):
        return all(_all_equal(e, val) for e in v)
    return v == val


# ---- parameter holders (MONAI attribute names; never called) ------------------------------------------------------------
class _Conv(nn.Module):
    """monai Convolution(conv_only=True): a Sequential whose only member is `conv`."""

    def __init__(self, conv):
        super().__init__()
        self.conv = conv


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = _Conv(nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False))
        self.conv2 = _Conv(nn.Conv3d(cout, cout, 3, stride=1, padding=1, bias=False))
        self.norm1 = nn.InstanceNorm3d(cout, affine=True)
        self.norm2 = nn.InstanceNorm3d(cout, affine=True)
        self.stride = stride


class _UpBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.transp_conv = _Conv(nn.ConvTranspose3d(cin, cout, 2, stride=2, bias=False))
        self.conv_block = _BasicBlock(2 * cout, cout, 1)


class _OutBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv(nn.Conv3d(cin, cout, 1, bias=True))


class _In:
    """A block input: Act + optional lazy prologue (per-(n,c) scale/shift, scalar or per-channel slope)."""
    __slots__ = ("act", "scale", "shift", "slope_vec")

    def __init__(self, act, scale=None, shift=None, slope_vec=None):
        self.act, self.scale, self.shift, self.slope_vec = act, scale, shift, slope_vec

    def kw(self):
        if self.scale is None:
            return dict(in_mode=IN_PLAIN)
        return dict(in_mode=IN_AFFINE_ACT, scale=self.scale, shift=self.shift, slope=SLOPE, in_slope=self.slope_vec)


def _tw_fwd(w):
    """ConvTranspose3d weight [ci, co, 2,2,2] -> 1x1x1 GEMM weight [(p, co), ci, 1,1,1], p = 4a+2b+e."""
    ci, co = w.shape[0], w.shape[1]
    return w.permute(2, 3, 4, 1, 0).reshape(8 * co, ci, 1, 1, 1)


class HipDynUNet(HipNetBase):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, strides, upsample_kernel_size, filters=None,
                 dropout=None, norm_name=("INSTANCE", {"affine": True}),
                 act_name=("leakyrelu", {"inplace": True, "negative_slope": 0.01}), deep_supervision=False, deep_supr_num=1,
                 res_block=False, trans_bias=False):
        super().__init__()
        bad = []
        if spatial_dims != 3:
            bad.append("spatial_dims != 3")
        if not _all_equal(kernel_size, 3):
            bad.append("kernel_size != 3")
        if not (_all_equal(strides[0], 1) and _all_equal(list(strides[1:]), 2)):
            bad.append("strides other than [1, 2, 2, ...]")
        if not _all_equal(upsample_kernel_size, 2) or len(upsample_kernel_size) != len(strides) - 1:
            bad.append("upsample_kernel_size other than 2 per level")
        if deep_supervision:
            bad.append("deep_supervision=True")
        if res_block:
            bad.append("res_block=True")
        if dropout is not None:
            bad.append("dropout")
        if trans_bias:
            bad.append("trans_bias=True")
        nn_ = norm_name[0] if isinstance(norm_name, (tuple, list)) else norm_name
        if str(nn_).lower() != "instance" or (isinstance(norm_name, (tuple, list)) and not norm_name[1].get("affine", False)):
            bad.append("norm other than InstanceNorm(affine=True)")
        an = act_name[0] if isinstance(act_name, (tuple, list)) else act_name
        if str(an).lower() != "leakyrelu" or (isinstance(act_name, (tuple, list)) and act_name[1].get("negative_slope", 0.01) != 0.01):
            bad.append("activation other than LeakyReLU(0.01)")
        if bad:
            raise NotImplementedError("HipDynUNet implements the configuration the reference ships (brats2020_config.json); "
                                      "unsupported: " + ", ".join(bad))
        L = len(strides)
        if filters is None:
            filters = [min(2 ** (5 + i), 320) for i in range(L)]     # MONAI default for spatial_dims == 3
        filters = list(filters)[:L]
        if len(filters) != L or any(f % 4 for f in filters) or out_channels > 8 or filters[0] > 96:
            raise NotImplementedError("filters must be multiples of 4, one per level; out_channels <= 8; filters[0] <= 96")
        self.spatial_dims, self.in_channels, self.out_channels = spatial_dims, in_channels, out_channels
        self.kernel_size, self.strides, self.upsample_kernel_size = kernel_size, strides, upsample_kernel_size
        self.filters = filters
        self.deep_supervision = False
        # registration order == MONAI's (state_dict key order)
        self.input_block = _BasicBlock(in_channels, filters[0], 1)
        self.downsamples = nn.ModuleList(_BasicBlock(filters[i - 1], filters[i], 2) for i in range(1, L - 1))
        self.bottleneck = _BasicBlock(filters[-2], filters[-1], 2)
        self.upsamples = nn.ModuleList(_UpBlock(filters[L - 1 - k], filters[L - 2 - k]) for k in range(L - 1))
        self.output_block = _OutBlock(filters[0], out_channels)
        self.deep_supervision_heads = nn.ModuleList()
        # DynUNet.initialize_weights: kaiming_normal_(a=0.01) on conv / transposed-conv weights, zero biases
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                nn.init.kaiming_normal_(m.weight, a=0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self.n_in_channels = in_channels
        self._init_engine()

    # MONAI registers the same blocks a second time under `skip_layers.*`: DynUNet.__init__ ends with
    #   self.skip_layers = create_skips(0, [input_block] + downsamples, upsamples[::-1], bottleneck)
    # a chain of DynUNetSkipLayer(downsample, next_layer, upsample) modules (attributes registered in that order) whose innermost
    # `next_layer` is the bottleneck block itself. nn.Module.state_dict() does not de-duplicate shared modules, so a MONAI checkpoint
    # carries every encoder / decoder tensor twice: under its own name and under the alias. Emitting the aliases (same tensors, MONAI's
    # order: after output_block) makes a HipDynUNet checkpoint load into MONAI's DynUNet with strict=True; loading accepts and ignores
    # them. [MONAI-memory: the class is not importable here; the key names follow SURVEY.md 8a-b5 / MONAI's dynunet.py.]
    def _skip_layer_aliases(self):
        L = len(self.filters)
        downs = [("input_block", self.input_block)] + [(f"downsamples.{i}", m) for i, m in enumerate(self.downsamples)]
        ups = [(f"upsamples.{k}", m) for k, m in enumerate(self.upsamples)][::-1]
        out = []

        def walk(prefix, depth):
            if depth == L - 1:
                out.append((prefix, "bottleneck"))
                return
            out.append((prefix + ".downsample", downs[depth][0]))
            walk(prefix + ".next_layer", depth + 1)
            out.append((prefix + ".upsample", ups[depth][0]))
        walk("skip_layers", 0)
        return out                                   # [(alias prefix, canonical prefix)]

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        own = [k for k in sd.keys() if k.startswith(prefix)]
        for alias, canon in self._skip_layer_aliases():
            for k in own:
                if k.startswith(prefix + canon + "."):
                    sd[prefix + alias + k[len(prefix + canon):]] = sd[k]
        return sd

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {k: v for k, v in state_dict.items() if not k.startswith("skip_layers.")}
        return super().load_state_dict(sd, strict=strict, **kw)

    def forward(self, x):
        return self._run(x)

    # ---- forward -------------------------------------------------------------------------------------------------------
    def _blocks(self):
        return [self.input_block] + list(self.downsamples) + [self.bottleneck]

    @staticmethod
    def _ci_pad(blk, cin):
        """Weight transform for a block whose input Act carries more channels than conv1 has (the network input, zero-padded to a
        multiple of 4 channels because NDHWC rows are float4 multiples): zero input-channel columns."""
        pad = cin - blk.conv1.conv.in_channels
        return None if pad == 0 else (lambda w: F.pad(w, (0, 0, 0, 0, 0, 0, 0, pad)))

    def _block_fwd(self, be, blk, xin, dst, keep):
        n = xin.act.shape[0]
        cout = blk.norm1.num_features
        d, h, w = dst.shape[1:4]
        r1 = be.empty_act(n, d, h, w, cout)
        # InstanceNorm statistics are not passes over r1 / dst: each conv's epilogue leaves the moment records of what it wrote
        # (csrc/gn_fuse.h) and gn_stats finalises those
        be.conv_fwd(xin.act, self._packed_weight(blk.conv1.conv.weight, 0, self._ci_pad(blk, xin.act.c)), r1, 3, blk.stride, moments=True,
                    **xin.kw())
        st1 = be.gn_stats(r1, cout, IN_EPS, blk.norm1.weight.data, blk.norm1.bias.data)
        r1.mom = None
        be.conv_fwd(r1, self._packed_weight(blk.conv2.conv.weight, 0), dst, 3, 1, in_mode=IN_AFFINE_ACT, scale=st1[1], shift=st1[2],
                    slope=SLOPE, moments=True)
        st2 = be.gn_stats(dst, cout, IN_EPS, blk.norm2.weight.data, blk.norm2.bias.data)
        dst.mom = None
        return dict(xin=xin, r1=r1, st1=st1, r2=dst, st2=st2) if keep else dict(r2=dst, st2=st2)

    def _forward_impl(self, x, keep):
        be = self._begin_forward()
        n, _, D, H, W = x.shape
        f, L = self.filters, len(self.filters)
        sizes = [(D, H, W)]
        for _ in range(L - 1):
            if any(s % 2 for s in sizes[-1]):
                raise ValueError(f"HipDynUNet: spatial size {tuple(x.shape[2:])} must be divisible by {2 ** (L - 1)} "
                                 "(ConvTranspose3d(k2,s2) output must match the skip, as in MONAI DynUNet)")
            sizes.append(tuple(s // 2 for s in sizes[-1]))
        cpad = (self.in_channels + 3) // 4 * 4
        if cpad == self.in_channels:
            xa = be.empty_act(n, D, H, W, self.in_channels)
            be.ncdhw_to_ndhwc(x, xa)
        else:
            xa = be.zeros_act(n, D, H, W, cpad)
            be.ncdhw_to_ndhwc(x, xa.slice(0, self.in_channels))
        cats = [be.empty_act(n, *sizes[i], 2 * f[i]) for i in range(L - 1)]
        enc = []
        cur = _In(xa)
        for i, blk in enumerate(self._blocks()):
            dst = cats[i].slice(f[i], f[i]) if i < L - 1 else be.empty_act(n, *sizes[i], f[i])
            s = self._block_fwd(be, blk, cur, dst, keep)
            enc.append(s)
            cur = _In(s["r2"], s["st2"][1], s["st2"][2])
        ups = []
        ones = zeros = None
        for k, up in enumerate(self.upsamples):
            lvl = L - 2 - k
            co = f[lvl]
            cat = cats[lvl]
            be.conv_fwd(cur.act, self._packed_weight(up.transp_conv.conv.weight, 0, _tw_fwd), cat.slice(0, co), 1, out_mode=OUT_D2S,
                        **cur.kw())
            sk = enc[lvl]["st2"]
            ones = torch.ones(n, co, dtype=torch.float32, device=be.device)
            zeros = torch.zeros(n, co, dtype=torch.float32, device=be.device)
            sc = torch.cat((ones, sk[1]), 1).contiguous()
            sh = torch.cat((zeros, sk[2]), 1).contiguous()
            sl = torch.cat((ones[0], torch.full((co,), SLOPE, dtype=torch.float32, device=be.device))).contiguous()
            cin_ = _In(cat, sc, sh, sl)
            dst = be.empty_act(n, *sizes[lvl], co)
            s = self._block_fwd(be, up.conv_block, cin_, dst, keep)
            s["tin"] = cur
            ups.append(s)
            cur = _In(s["r2"], s["st2"][1], s["st2"][2])
        logits = torch.empty(n, self.out_channels, D, H, W, dtype=torch.float32, device=x.device)
        ob = self.output_block.conv.conv
        be.proj_fwd(cur.act, ob.weight.data.reshape(self.out_channels, -1), ob.bias.data, logits, cur.scale, cur.shift, SLOPE)
        self._end_forward()
        saved = dict(enc=enc, ups=ups, sizes=sizes, n=n, last=cur) if keep else None
        return logits, saved

    # ---- backward ------------------------------------------------------------------------------------------------------
    def _block_bwd(self, be, blk, s, dA2, need_dx, dx_residual=None):
        """dA2: gradient wrt the block's ACTIVATED output (modified in place). Returns the gradient wrt the block's ACTIVATED
        input (Act) or None."""
        cout = blk.norm1.num_features
        r1, r2, st1, st2, xin = s["r1"], s["r2"], s["st1"], s["st2"], s["xin"]
        be.gn_act_bwd(r2, dA2, dA2, cout, SLOPE, blk.norm2.weight.data, st2[0], st2[1], st2[2],
                      self._gslice(blk.norm2.weight), self._gslice(blk.norm2.bias))
        d_r2 = dA2
        with self._wgrad_stream(be, r1, d_r2, st1[1], st1[2]):
            be.conv_wgrad(r1, d_r2, self._gslice(blk.conv2.conv.weight), 3, 1, in_mode=IN_AFFINE_ACT, scale=st1[1], shift=st1[2], slope=SLOPE)
        dA1 = be.empty_act(*r1.shape)
        # the dgrad's epilogue also leaves the first pass of the norm backward (sum du, sum du*xhat per tile)
        p1 = be.conv_fwd(d_r2, self._packed_weight(blk.conv2.conv.weight, 1), dA1, 3, 1, gnb=(r1, st1, cout, SLOPE))
        be.gn_act_bwd(r1, dA1, dA1, cout, SLOPE, blk.norm1.weight.data, st1[0], st1[1], st1[2],
                      self._gslice(blk.norm1.weight), self._gslice(blk.norm1.bias), partials=p1)
        d_r1 = dA1
        w1 = blk.conv1.conv.weight
        with self._wgrad_stream(be, xin.act, d_r1, xin.scale, xin.shift, xin.slope_vec):
            if xin.act.c == w1.shape[1]:
                be.conv_wgrad(xin.act, d_r1, self._gslice(w1), 3, blk.stride, **xin.kw())
            else:                                  # zero-padded network input: wgrad over the padded channels, sliced back
                tw = torch.empty(w1.shape[0], xin.act.c, 3, 3, 3, dtype=torch.float32, device=w1.device)
                be.conv_wgrad(xin.act, d_r1, tw, 3, blk.stride, **xin.kw())
                self._gslice(w1).copy_(tw[:, :w1.shape[1]])
        self._flush_ready()
        if not need_dx:
            return None
        n, d, h, w, cin = xin.act.shape
        dAin = be.empty_act(n, d, h, w, cin)
        wp = self._packed_weight(blk.conv1.conv.weight, 1, self._ci_pad(blk, cin))
        if blk.stride == 1:
            be.conv_fwd(d_r1, wp, dAin, 3, 1, residual=dx_residual)
        else:
            be.conv_fwd(d_r1, wp, dAin, 3, 1, pad=1, in_mode=IN_ZERO_INSERT, residual=dx_residual, out_dhw=(d, h, w))
        return dAin

    def _backward_impl_body(self, be, saved, dlogits, need_dx):
        f, L = self.filters, len(self.filters)
        enc, ups, n = saved["enc"], saved["ups"], saved["n"]
        last = saved["last"]
        ob = self.output_block.conv.conv
        dA = be.empty_act(*last.act.shape)
        be.proj_bwd(last.act, ob.weight.data.reshape(self.out_channels, -1), dlogits, dA,
                    self._gslice(ob.weight).view(self.out_channels, -1), self._gslice(ob.bias), last.scale, last.shift, SLOPE)
        d_skip = [None] * (L - 1)
        for k in range(L - 2, -1, -1):
            lvl = L - 2 - k
            co = f[lvl]
            up = self.upsamples[k]
            s = ups[k]
            dcat = self._block_bwd(be, up.conv_block, s, dA, True)
            d_up, d_skip[lvl] = dcat.slice(0, co), dcat.slice(co, co)
            tin = s["tin"]                                  # activated input of the transposed conv (raw + prologue)
            w = up.transp_conv.conv.weight
            cin = w.shape[0]
            with self._wgrad_stream(be, tin.act, d_up, tin.scale, tin.shift, tin.slope_vec):
                dw1 = torch.empty(8 * co, cin, dtype=torch.float32, device=be.device)
                be.conv_wgrad(tin.act, d_up, dw1, 1, out_mode=OUT_D2S, **tin.kw())
                self._gslice(w).copy_(dw1.view(2, 2, 2, co, cin).permute(4, 3, 0, 1, 2))
            dA = be.empty_act(*tin.act.shape)
            be.conv_fwd(d_up, self._packed_weight(w, 1, _tw_fwd), dA, 1, in_mode=IN_S2D)
            self._flush_ready()
        blocks = self._blocks()
        dx = None
        for i in range(L - 1, -1, -1):
            need = need_dx or i > 0
            dA = self._block_bwd(be, blocks[i], enc[i], dA, need, d_skip[i - 1] if i > 0 else None)
            if i == 0:
                dx = dA
        dx_t = None
        if need_dx and dx is not None:
            sizes = saved["sizes"]
            dx_t = torch.empty(n, self.in_channels, *sizes[0], dtype=torch.float32, device=dlogits.device)
            be.ndhwc_to_ncdhw(dx.slice(0, self.in_channels), dx_t)
        return dx_t
