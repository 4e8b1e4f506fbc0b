"""HipUNet3D: drop-in for the reference's UNet3D (unet3d/models/pytorch/segmentation/unet.py:47-50) on MI355X.

Same constructor kwargs as the reference's config surface (autoencoder/variational.py:37-42), same parameter names,
shapes (OIDHW) and default initialisation order (so the same torch seed gives the same weights and checkpoints
interchange), NCDHW fp32 in / NCDHW fp32 logits out, differentiable. The arithmetic is
entirely the HIP library (include/mi355_unet3d.h): the holder modules below only own parameters and are never called.

Forward graph (reference lines in brackets), all activations NDHWC:
  encoder level i  : residual blocks [myronenko.py:47-58] = gn_stats -> conv3(GN+ReLU prologue) -> gn_stats ->
                     conv3(GN+ReLU prologue, + identity | 1x1 shortcut, * Dropout3d mask) ; stride-2 conv3 down
                     [myronenko.py:104-105, unet.py:8-16]. The level output is written straight into the channel
                     slice of the decoder's concat buffer (no torch.cat copy).
  decoder level    : block -> 1x1 conv -> trilinear x2 (or ConvTranspose3d k3 s2 p1) -> pad/crop window -> concat
                     [unet.py:27-44, decoder.py:99-106].
  head             : 1x1 projection to n_outputs written as NCDHW [variational.py:59-60, unet.py:50].
Backward is explicit (no autograd graph inside): wgrad / dgrad (same MFMA conv kernel on flipped packs) /
GroupNorm+ReLU backward / transposed upsample, parameter gradients written into one flat buffer.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .engine import HipNetBase
from ._lib import IN_AFFINE_ACT, IN_ZERO_INSERT

GN_EPS = 1e-5


def _groups(planes, norm_groups=8):
    """GroupNorm group count of the reference block (myronenko.py:23-31)."""
    if planes < norm_groups or planes % norm_groups:
        return planes
    return norm_groups


# ---- parameter holders (names == reference attribute names; never called) --------------------------------------
class _ConvBlock(nn.Module):
    def __init__(self, cin, cout, stride=1, kernel_size=3):
        super().__init__()
        self.norm1 = nn.GroupNorm(_groups(cin), cin)
        self.conv = nn.Conv3d(cin, cout, kernel_size, stride=stride, padding=kernel_size // 2, bias=False)


class _ResBlock(nn.Module):
    def __init__(self, cin, cout, kernel_size=3):
        super().__init__()
        self.conv1 = _ConvBlock(cin, cout, kernel_size=kernel_size)
        self.conv2 = _ConvBlock(cout, cout, kernel_size=kernel_size)
        self.sample = nn.Conv3d(cin, cout, 1, bias=False) if cin != cout else None


class _Layer(nn.Module):
    def __init__(self, n_blocks, cin, cout, dropout=None, kernel_size=3):
        super().__init__()
        self.blocks = nn.ModuleList()
        for _ in range(n_blocks):
            self.blocks.append(_ResBlock(cin, cout, kernel_size))
            cin = cout
        self.dropout_p = dropout


class _Encoder(nn.Module):
    def __init__(self, n_features, base_width, layer_blocks, feature_dilation, downsampling_stride, dropout=0.2, kernel_size=3):
        super().__init__()
        self.layers = nn.ModuleList()
        self.downsampling_convolutions = nn.ModuleList()
        self.widths = []
        cin = n_features
        for i, nb in enumerate(layer_blocks):
            cout = base_width * (feature_dilation ** i)
            self.layers.append(_Layer(nb, cin, cout, dropout if (dropout and i == 0) else None, kernel_size))
            if i != len(layer_blocks) - 1:
                self.downsampling_convolutions.append(
                    nn.Conv3d(cout, cout, kernel_size, stride=downsampling_stride, padding=kernel_size // 2, bias=False))
            self.widths.append(cout)
            cin = cout


class _Decoder(nn.Module):
    def __init__(self, base_width, layer_blocks, feature_reduction_scale, use_transposed_convolutions, kernel_size=3):
        super().__init__()
        self.layers = nn.ModuleList()
        self.pre_upsampling_blocks = nn.ModuleList()
        self.upsampling_blocks = nn.ModuleList() if use_transposed_convolutions else []
        self.level_widths = []
        nl = len(layer_blocks)
        for i, nb in enumerate(layer_blocks):
            depth = nl - (i + 1)
            # MirroredDecoder.calculate_layer_widths (decoder.py:111-122) + UNetDecoder override (unet.py:20-25)
            if depth > 0:
                out_w = int(base_width * (feature_reduction_scale ** (depth - 1)))
                in_w = out_w * feature_reduction_scale
            else:
                out_w = in_w = base_width
            if depth != nl - 1:
                in_w *= 2
            self.level_widths.append((in_w, out_w))
            if depth != 0:
                self.layers.append(_Layer(nb, in_w, in_w, None, kernel_size))
                if use_transposed_convolutions:
                    self.pre_upsampling_blocks.append(nn.Sequential())
                    self.upsampling_blocks.append(nn.ConvTranspose3d(in_w, out_w, kernel_size, stride=2, padding=1))
                else:
                    self.pre_upsampling_blocks.append(nn.Conv3d(in_w, out_w, 1, bias=False))
            else:
                self.layers.append(_Layer(nb, in_w, out_w, None, kernel_size))


class _Saved:
    """Activations and statistics a residual block keeps for its backward."""
    __slots__ = ("x", "h1", "st1", "st2", "out", "chscale", "x_lp")


class HipUNet3D(HipNetBase):
    def __init__(self, input_shape=None, n_features=1, base_width=32, encoder_blocks=None, decoder_blocks=None,
                 feature_dilation=2, downsampling_stride=2, interpolation_mode="trilinear", n_outputs=1, layer_widths=None,
                 decoder_mirrors_encoder=False, activation=None, use_transposed_convolutions=False, kernel_size=3):
        super().__init__()
        if layer_widths is not None:
            # unusable in the reference as well (IndexError at decoder.py:112-114)
            raise ValueError("layer_widths is not supported (the reference UNet3D raises IndexError for it)")
        if kernel_size != 3:
            # the reference forwards kernel_size to every conv3x3x3 and to ConvTranspose3d (myronenko.py:15, decoder.py:96-102); the HIP
            # kernels exist for the 3x3x3 form every shipped configuration uses ("constructor surface")
            raise NotImplementedError("HipUNet3D implements the reference default kernel_size=3")
        if downsampling_stride != 2:
            # the reference uses it as conv stride AND as the decoder's up-sampling scale (variational.py:47-52)
            raise NotImplementedError("HipUNet3D implements the reference default downsampling_stride=2")
        # interpolation_mode: the reference accepts any string at construction and hands it to F.interpolate together with
        # align_corners=False (decoder.py:105-106); torch then raises at the first forward for everything but "trilinear"
        # ("nearest" / "area": ValueError, align_corners cannot be set; "linear" / "bilinear" / "bicubic": NotImplementedError for
        # 5-D input). Same here: accepted now, the same exception from the first forward (see forward()).
        self.interpolation_mode = interpolation_mode
        # NDHWC rows are float4 multiples: an input with 1..3 (5..7, ...) channels is carried with zero channels up to the next
        # multiple of 4; the first block's weights / norm parameters are zero-padded copies and their gradients are sliced back
        self._cin_pad = (n_features + 3) // 4 * 4
        self.input_shape = input_shape
        self.base_width = base_width
        self.n_features = n_features
        self.n_outputs = n_outputs
        self.use_transposed_convolutions = use_transposed_convolutions
        if encoder_blocks is None:
            encoder_blocks = [1, 2, 2, 4]
        if decoder_mirrors_encoder:
            decoder_blocks = encoder_blocks
        elif decoder_blocks is None:
            decoder_blocks = [1] * len(encoder_blocks)
        if len(decoder_blocks) != len(encoder_blocks):
            raise ValueError("decoder_blocks must have one entry per encoder level")
        self.encoder = _Encoder(n_features, base_width, encoder_blocks, feature_dilation, downsampling_stride, kernel_size=kernel_size)
        self.decoder = _Decoder(base_width, decoder_blocks, feature_dilation, use_transposed_convolutions, kernel_size)
        # the reference draws a throw-away final conv (variational.py:54) before UNet3D replaces it (unet.py:50):
        # reproduce the RNG consumption so the same seed yields the same weights.
        nn.Conv3d(base_width, n_features, 1, bias=False)
        self.final_convolution = nn.Conv3d(base_width, n_outputs, 1, bias=False)
        if activation == "sigmoid":
            self.activation = nn.Sigmoid()
        elif activation == "softmax":
            self.activation = nn.Softmax(dim=1)
        else:
            self.activation = None
        self.n_in_channels = n_features
        self.dropout_generator = None
        self.last_dropout_scale = None
        self._init_engine()

    # ---- forward -----------------------------------------------------------------------------------------------
    def forward(self, x):
        if self.interpolation_mode != "trilinear" and not self.use_transposed_convolutions:
            # let torch raise exactly what the reference's F.interpolate call raises for this mode (decoder.py:105-106)
            F.interpolate(torch.zeros(1, 1, 1, 1, 1), scale_factor=2, mode=self.interpolation_mode, align_corners=False)
            raise NotImplementedError(f"interpolation_mode={self.interpolation_mode!r}: only 'trilinear' has a HIP kernel")
        y = self._run(x)
        if self.activation is not None:
            y = self.activation(y)
        return y

    def _block_fwd(self, be, blk, x, out, chscale, keep, out_moments=True, x_lp=None):
        """One residual block. x: Act input; out: Act destination (maybe a concat slice). Norm statistics are not passes over the
        tensors: each conv's epilogue leaves the moment records of what it wrote (x.mom / h1.mom, csrc/gn_fuse.h) and gn_stats
        finalises those; `out_moments`: whether `out` is normalised by whoever consumes it."""
        n, d, h, w, cin = x.shape
        cout = blk.conv1.conv.out_channels
        c1, c2 = blk.conv1, blk.conv2
        g1, b1, groups1, padw = self._in_pad(c1, cin)
        st1 = be.gn_stats(x, groups1, GN_EPS, g1, b1)
        h1 = be.empty_act(n, d, h, w, cout)
        be.conv_fwd(x, self._packed_weight(c1.conv.weight, 0, padw), h1, 3, 1, in_mode=IN_AFFINE_ACT, scale=st1[1], shift=st1[2],
                    moments=True)
        st2 = be.gn_stats(h1, c2.norm1.num_groups, GN_EPS, c2.norm1.weight.data, c2.norm1.bias.data)
        h1.mom = None
        if blk.sample is not None:
            idn = be.empty_act(n, d, h, w, cout)
            # x_lp: the 16-bit copy of an fp32 block input (the network input under 16-bit activation storage): what a conv reads of it
            be.conv_fwd(x if x_lp is None else x_lp, self._packed_weight(blk.sample.weight, 0, padw), idn, 1)
        else:
            idn = x
        be.conv_fwd(h1, self._packed_weight(c2.conv.weight, 0), out, 3, 1, in_mode=IN_AFFINE_ACT, scale=st2[1], shift=st2[2],
                    residual=idn, chscale=chscale, moments=out_moments)
        if keep:
            s = _Saved()
            s.x, s.h1, s.st1, s.st2, s.out, s.chscale, s.x_lp = x, h1, st1, st2, out, chscale, x_lp
            return s
        return None

    def _in_pad(self, c1, cin):
        """(gamma, beta, groups, weight transform) of a block whose input Act carries `cin` >= conv.in_channels channels (the
        zero-padded network input). Padding only happens for channel counts that are not multiples of 4, where the reference's
        GroupNorm is per-channel (groups = C whenever C < 8 or C % 8 != 0, myronenko.py:23-31): groups = cin keeps it so, and
        the zero gamma / beta / weight columns make the pad channels contribute nothing."""
        cm = c1.conv.in_channels
        if cin == cm:
            return c1.norm1.weight.data, c1.norm1.bias.data, c1.norm1.num_groups, None
        pad = cin - cm
        return (F.pad(c1.norm1.weight.data, (0, pad)), F.pad(c1.norm1.bias.data, (0, pad)), cin,
                lambda w: F.pad(w, (0, 0, 0, 0, 0, 0, 0, pad)))

    def _wgrad_target(self, p, cin):
        """Destination of a weight gradient whose kernel sees `cin` input channels: the parameter's slice of the flat gradient
        buffer, or (padded input) a scratch tensor that `_wgrad_commit` slices back into it."""
        if p.shape[1] == cin:
            return self._gslice(p)
        return torch.empty(p.shape[0], cin, *p.shape[2:], dtype=torch.float32, device=p.device)

    def _wgrad_commit(self, p, t):
        if t.shape != p.shape:
            self._gslice(p).copy_(t[:, :p.shape[1]])

    def _layer_fwd(self, be, layer, x, out_last, keep, out_moments=True, x_lp=None):
        """All blocks of a layer; the last block writes into out_last (`out_moments`: a norm reads it next). Returns list of saved blocks."""
        saved = []
        nb = len(layer.blocks)
        n, d, h, w, _ = x.shape
        for j, blk in enumerate(layer.blocks):
            cout = blk.conv1.conv.out_channels
            out = out_last if j == nb - 1 else be.empty_act(n, d, h, w, cout)
            chscale = None
            if j == 0 and layer.dropout_p and self.training:
                # Dropout3d(p) after block 0 (myronenko.py:75-80): per-(n, channel) Bernoulli keep mask / (1-p)
                p = layer.dropout_p
                keepmask = torch.rand(n, cout, device=be.device, generator=self.dropout_generator) >= p
                chscale = keepmask.float() / (1.0 - p)
                self.last_dropout_scale = chscale        # [N, C] keep mask / (1 - p) the forward drew (what parity tests hand the oracle)
            saved.append(self._block_fwd(be, blk, x, out, chscale, keep, out_moments or j < nb - 1, x_lp if j == 0 else None))
            x = out
        return saved

    def _forward_impl(self, x, keep):
        be = self._begin_forward()
        n, _, D, H, W = x.shape
        enc, dec = self.encoder, self.decoder
        L = len(enc.layers)
        # spatial sizes per level (stride-2 k3 p1 conv: ceil(n/2))
        sizes = [(D, H, W)]
        for _ in range(L - 1):
            sizes.append(tuple((s - 1) // 2 + 1 for s in sizes[-1]))
        # The network input stays fp32 when it has the first-layer kernels' 4 channels (its norm statistics, the normalising conv and that
        # conv's weight gradient read it as the reference's GroupNorm does: fp32); under 16-bit activation storage the 1x1x1 shortcut of the
        # first block reads a 16-bit copy (what autocast's conv does with an fp32 input). Wider inputs are stored like every other activation.
        xa_dtype = torch.float32 if self._cin_pad == 4 else be.act_dtype
        if self._cin_pad == self.n_features:
            xa = be.empty_act(n, D, H, W, self.n_features, dtype=xa_dtype)
            be.ncdhw_to_ndhwc(x, xa)
        else:
            xa = be.zeros_act(n, D, H, W, self._cin_pad, dtype=xa_dtype)
            be.ncdhw_to_ndhwc(x, xa.slice(0, self.n_features))
        xa_lp = be.cast(xa, be.act_dtype) if xa.dtype != be.act_dtype else None
        # concat buffers for decoder levels: level i (resolution of encoder level i, i < L-1) holds [up | skip]
        cats = []
        for i in range(L - 1):
            up_c = dec.level_widths[L - 2 - i][1]      # out width of the decoder layer feeding this level
            skip_c = enc.widths[i]
            d_, h_, w_ = sizes[i]
            cats.append((be.empty_act(n, d_, h_, w_, up_c + skip_c), up_c, skip_c))
        saved = {"enc": [], "dec": [], "sizes": sizes, "xa": xa, "cats": cats, "n": n}
        cur = xa
        enc_out = []
        for i, layer in enumerate(enc.layers):
            d_, h_, w_ = sizes[i]
            if i < L - 1:
                cat, up_c, skip_c = cats[i]
                out = cat.slice(up_c, skip_c)
            else:
                out = be.empty_act(n, d_, h_, w_, enc.widths[i])
            saved["enc"].append(self._layer_fwd(be, layer, cur, out, keep, x_lp=xa_lp if i == 0 else None))
            enc_out.append(out)
            if i < L - 1:
                dn = sizes[i + 1]
                nxt = be.empty_act(n, dn[0], dn[1], dn[2], enc.widths[i])
                be.conv_fwd(out, self._packed_weight(enc.downsampling_convolutions[i].weight, 0), nxt, 3, 2, moments=True)
                cur = nxt
        # decoder
        cur = enc_out[-1]
        for k in range(L - 1):
            layer = dec.layers[k]
            lvl = L - 2 - k                      # resolution level this decoder stage upsamples TO
            in_w, out_w = dec.level_widths[k]
            d_, h_, w_ = sizes[lvl + 1]
            lay_out = be.empty_act(n, d_, h_, w_, in_w)
            sv = self._layer_fwd(be, layer, cur, lay_out, keep, out_moments=False)     # feeds the 1x1x1 / transposed conv, not a norm
            cat, up_c, skip_c = cats[lvl]
            tgt = sizes[lvl]
            off = tuple((t - 2 * s) // 2 for t, s in zip(tgt, (d_, h_, w_)))
            if self.use_transposed_convolutions:
                up = dec.upsampling_blocks[k]
                off = tuple((t - (2 * s - 1)) // 2 for t, s in zip(tgt, (d_, h_, w_)))
                cat.tensor()[..., :up_c].zero_()
                be.conv_fwd(lay_out, self._packed_weight(up.weight, 2), cat.slice(0, up_c), 3, 1, pad=1, in_mode=IN_ZERO_INSERT,
                            bias=up.bias.data, off=off, out_dhw=tuple(2 * s - 1 for s in (d_, h_, w_)))
                pre_out = None
            else:
                pre_out = be.empty_act(n, d_, h_, w_, out_w)
                be.conv_fwd(lay_out, self._packed_weight(dec.pre_upsampling_blocks[k].weight, 0), pre_out, 1)
                be.upsample2x_fwd(pre_out, cat.slice(0, up_c), off)
            # statistics of the concat buffer: the skip half's records were left by the encoder conv that wrote it, the up-sampled
            # half (no conv epilogue produced it) gets one read of that half only
            skip_mom = enc_out[lvl].mom
            if be.fused_stats and skip_mom is not None:
                upsl = cat.slice(0, up_c)
                cat.mom = be.moments(upsl) + skip_mom
            saved["dec"].append((sv, lay_out, off))
            cur = cat
        d_, h_, w_ = sizes[0]
        last_out = be.empty_act(n, d_, h_, w_, dec.level_widths[-1][1])
        saved["last"] = self._layer_fwd(be, dec.layers[-1], cur, last_out, keep, out_moments=False)
        saved["last_out"] = last_out
        logits = torch.empty(n, self.n_outputs, D, H, W, dtype=torch.float32, device=x.device)
        wf = self.final_convolution.weight
        be.proj_fwd(last_out, wf.data.reshape(self.n_outputs, -1), None, logits)
        self._end_forward()
        return logits, (saved if keep else None)

    # ---- backward ----------------------------------------------------------------------------------------------
    def _block_bwd(self, be, blk, s, d_out, need_dx, dx_out=None):
        """d_out: Act gradient wrt the block output (modified in place by the dropout scale). Returns Act dx or None."""
        c1, c2 = blk.conv1, blk.conv2
        n, d, h, w, cin = s.x.shape
        cout = c1.conv.out_channels
        if s.chscale is not None:
            be.chscale(d_out, s.chscale, d_out)
        st1, st2 = s.st1, s.st2
        # Weight gradients go to the side stream (engine._wgrad_stream), each forked BEFORE the data gradient conv that follows it in
        # program order. Measured and rejected in round 4 (profiles/r4_ab_experiments.txt): forking it AFTER that conv, so that the side
        # stream runs one matrix-bound kernel behind and the HBM-bound norm-backward kernels find a weight gradient in flight beside
        # them: 55.64 -> 56.27 ms per fp32 step, bf16 / C3 unchanged (the two streams' matrix-bound kernels share the CUs either way and
        # the lagging stream drains alone at the end of backward).
        with self._wgrad_stream(be, s.h1, d_out, st2[1], st2[2]):
            be.conv_wgrad(s.h1, d_out, self._gslice(c2.conv.weight), 3, 1, in_mode=IN_AFFINE_ACT, scale=st2[1], shift=st2[2])
        dA2 = be.empty_act(n, d, h, w, cout)
        # the dgrad's epilogue also leaves the first pass of the norm backward (sum du, sum du*xhat per tile): dA2 is in registers there
        p2 = be.conv_fwd(d_out, self._packed_weight(c2.conv.weight, 1), dA2, 3, 1, gnb=(s.h1, st2, c2.norm1.num_groups, 0.0))
        be.gn_act_bwd(s.h1, dA2, dA2, c2.norm1.num_groups, 0.0, c2.norm1.weight.data, st2[0], st2[1], st2[2],
                      self._gslice(c2.norm1.weight), self._gslice(c2.norm1.bias), partials=p2)
        dh1 = dA2
        g1, _, groups1, padw = self._in_pad(c1, cin)
        if (not need_dx and padw is None and blk.sample is not None
                and be.c4_bwd_supported(s.x, dh1, IN_AFFINE_ACT, 0.0, st1[1], st1[2])):
            # the network's first block (4 input channels, no gradient wrt the input): weight gradient of conv1 and dgamma / dbeta of
            # norm1 in ONE pass over dh1 -- no data-gradient tensor, no second read of dh1 (csrc/conv3d_c4_bwd.hip)
            be.c4_bwd(s.x, dh1, self._packed_weight(c1.conv.weight, 1), self._gslice(c1.conv.weight), groups1, g1, st1[0], st1[1], st1[2],
                      self._gslice(c1.norm1.weight), self._gslice(c1.norm1.bias))
            xs = s.x if s.x_lp is None else s.x_lp         # (16-bit storage: the shortcut conv read a 16-bit copy of the input)
            with self._wgrad_stream(be, xs, d_out):
                be.conv_wgrad(xs, d_out, self._gslice(blk.sample.weight), 1)
            self._flush_ready()
            return None
        with self._wgrad_stream(be, s.x, dh1, st1[1], st1[2]):
            tw = self._wgrad_target(c1.conv.weight, cin)
            be.conv_wgrad(s.x, dh1, tw, 3, 1, in_mode=IN_AFFINE_ACT, scale=st1[1], shift=st1[2])
            self._wgrad_commit(c1.conv.weight, tw)
        dA1 = be.empty_act(n, d, h, w, cin, dtype=s.x.dtype)          # a tensor's gradient is stored like the tensor
        p1 = be.conv_fwd(dh1, self._packed_weight(c1.conv.weight, 1, padw), dA1, 3, 1, gnb=(s.x, st1, groups1, 0.0))
        if blk.sample is not None:
            xs = s.x if s.x_lp is None else s.x_lp
            with self._wgrad_stream(be, xs, d_out):
                ts = self._wgrad_target(blk.sample.weight, cin)
                be.conv_wgrad(xs, d_out, ts, 1)
                self._wgrad_commit(blk.sample.weight, ts)
            d_id = None
            if need_dx:
                d_id = be.empty_act(n, d, h, w, cin)
                be.conv_fwd(d_out, self._packed_weight(blk.sample.weight, 1, padw), d_id, 1)
                if d_id.dtype != s.x.dtype:
                    d_id = be.cast(d_id, s.x.dtype)
        else:
            d_id = d_out
        dx = None
        if need_dx:
            dx = dx_out if dx_out is not None else dA1
        if padw is None:
            dg, db = self._gslice(c1.norm1.weight), self._gslice(c1.norm1.bias)
        else:
            dg, db = torch.empty(cin, dtype=torch.float32, device=be.device), torch.empty(cin, dtype=torch.float32, device=be.device)
        be.gn_act_bwd(s.x, dA1, dx if dx is not None else dA1, groups1, 0.0, g1, st1[0], st1[1], st1[2], dg, db,
                      addend=d_id if need_dx else None, partials=p1)
        if padw is not None:
            cm = c1.conv.in_channels
            self._gslice(c1.norm1.weight).copy_(dg[:cm])
            self._gslice(c1.norm1.bias).copy_(db[:cm])
        self._flush_ready()
        return dx

    def _layer_bwd(self, be, layer, saved, d_out, need_dx):
        for j in range(len(layer.blocks) - 1, -1, -1):
            d_out = self._block_bwd(be, layer.blocks[j], saved[j], d_out, need_dx or j > 0)
        return d_out

    def _backward_impl_body(self, be, saved, dlogits, need_dx):
        enc, dec = self.encoder, self.decoder
        L = len(enc.layers)
        sizes, cats, n = saved["sizes"], saved["cats"], saved["n"]
        # head
        last_out = saved["last_out"]
        wf = self.final_convolution.weight
        d_last = be.empty_act(*last_out.shape)
        be.proj_bwd(last_out, wf.data.reshape(self.n_outputs, -1), dlogits, d_last, self._gslice(wf).view(self.n_outputs, -1), None)
        d_cur = self._layer_bwd(be, dec.layers[-1], saved["last"], d_last, True)   # gradient wrt cats[0] (full width)
        # decoder, shallow -> deep
        d_skips = [None] * (L - 1)
        for k in range(L - 2, -1, -1):
            lvl = L - 2 - k
            cat, up_c, skip_c = cats[lvl]
            sv, lay_out, off = saved["dec"][k]
            d_cat = d_cur                                   # Act with cat's full channel count
            d_skips[lvl] = d_cat.slice(up_c, skip_c)
            in_w, out_w = dec.level_widths[k]
            d_lay = be.empty_act(*lay_out.shape)
            if self.use_transposed_convolutions:
                up = dec.upsampling_blocks[k]
                d_up = d_cat.slice(0, up_c)
                # ConvTranspose3d backward: dgrad = stride-2 correlation of the (window of the) output gradient
                dyw, dshape = self._window(be, d_up, off, tuple(2 * s - 1 for s in lay_out.shape[1:4]))
                be.conv_fwd(dyw, self._packed_weight(up.weight, 3), d_lay, 3, 2, pad=1)
                be.conv_wgrad(dyw, lay_out, self._gslice(up.weight), 3, 2, pad=1)
                self._gslice(up.bias).copy_(dyw.tensor().sum(dim=(0, 1, 2, 3), dtype=torch.float32))
            else:
                pre = dec.pre_upsampling_blocks[k]
                d_pre = be.empty_act(lay_out.shape[0], lay_out.shape[1], lay_out.shape[2], lay_out.shape[3], out_w)
                be.upsample2x_bwd(d_cat.slice(0, up_c), d_pre, off)
                with self._wgrad_stream(be, lay_out, d_pre):
                    be.conv_wgrad(lay_out, d_pre, self._gslice(pre.weight), 1)
                be.conv_fwd(d_pre, self._packed_weight(pre.weight, 1), d_lay, 1)
            d_cur = self._layer_bwd(be, dec.layers[k], sv, d_lay, True)
        # encoder, deep -> shallow; d_cur is the gradient wrt the deepest encoder output
        dx = None
        for i in range(L - 1, -1, -1):
            need = need_dx or i > 0
            d_in = self._layer_bwd(be, enc.layers[i], saved["enc"][i], d_cur, need)
            if i > 0:
                # gradient wrt encoder level i-1 output = dgrad of the stride-2 conv (+ the skip gradient from the concat)
                dw = enc.downsampling_convolutions[i - 1].weight
                out_prev = saved["enc"][i - 1][-1].out
                with self._wgrad_stream(be, out_prev, d_in):
                    be.conv_wgrad(out_prev, d_in, self._gslice(dw), 3, 2)
                dprev = be.empty_act(n, *sizes[i - 1], enc.widths[i - 1])
                be.conv_fwd(d_in, self._packed_weight(dw, 1), dprev, 3, 1, pad=1, in_mode=IN_ZERO_INSERT,
                            residual=d_skips[i - 1], out_dhw=sizes[i - 1])
                d_cur = dprev
            else:
                dx = d_in
        dx_t = None
        if need_dx and dx is not None:
            dx_t = torch.empty(n, self.n_features, *sizes[0], dtype=torch.float32, device=dlogits.device)
            be.ndhwc_to_ncdhw(dx.slice(0, self.n_features), dx_t)
        return dx_t

    def _window(self, be, d_up, off, win):
        """Gradient restricted to the F.pad window (unet.py:34-40) as a dense Act of extent `win`."""
        t = d_up.tensor()
        sl = []
        for o, wlen, full in zip(off, win, t.shape[1:4]):
            lo, hi = max(o, 0), min(o + wlen, full)
            sl.append(slice(lo, hi))
        sub = t[:, sl[0], sl[1], sl[2], :]
        n = t.shape[0]
        out = be.zeros_act(n, win[0], win[1], win[2], d_up.c)
        dst = out.tensor()[:, max(-off[0], 0):max(-off[0], 0) + sub.shape[1], max(-off[1], 0):max(-off[1], 0) + sub.shape[2],
                           max(-off[2], 0):max(-off[2], 0) + sub.shape[3], :]
        dst.copy_(sub)
        return out, win


class HipAutocastUNet(HipUNet3D):
    """Drop-in for the reference's AutocastUNet (unet3d/models/pytorch/segmentation/unet.py:53-58), which runs UNet3D.forward
    under torch.cuda.amp.autocast (fp16 convolutions with fp32 accumulate, norms in fp32). MI355X equivalent: the 3x3x3
    convolutions take the 16-bit matrix path (operands rounded while staged, fp32 accumulate); norm statistics, weights, weight
    gradients, logits and the loss stay fp32.
      autocast_dtype="bf16" (default): BASELINE configs[2]'s "bf16 mixed precision"; bf16 has fp32's exponent range, no GradScaler needed.
      autocast_dtype="fp16": the reference class's own arithmetic (CUDA autocast defaults to fp16; v_mfma_f32_32x32x16_f16): 8x smaller
        rounding error than bf16, fp16's range -- train it with torch's GradScaler as the reference's `training.amp` path does
        (train/training_utils.py:60-69, 93-96), which this module's backward supports (tests/test_boundary.py).
      activation_storage: "fp32" -- every activation tensor in HBM is fp32 and only the MFMA operands are rounded (the round-2..4 form) |
        "bf16" (with autocast_dtype="bf16") / "fp16" (with "fp16") -- conv outputs, block outputs, concat buffers and all their gradients
        are STORED in the 16-bit type of the mode, as the reference's autocast keeps conv outputs: half the bytes of
        every HBM-bound kernel of the step and of the saved-for-backward set; the network input (4 channels) stays fp32, values are
        rounded once, when stored. "fp16" is the reference's own amp form (fp16 tensors, train/train.py:33-37): like there, gradients
        stored as fp16 need the loss scale of torch's GradScaler (training_utils.py:60-69) -- unscaled Dice gradients sit in fp16's
        subnormal range. Default: "bf16" with autocast_dtype="bf16"; "fp32" with "fp16" (opt in to "fp16" together with a GradScaler)."""

    def __init__(self, *args, autocast_dtype="bf16", activation_storage=None, **kwargs):
        super().__init__(*args, **kwargs)
        name = {torch.bfloat16: "bf16", torch.float16: "fp16", torch.half: "fp16"}.get(autocast_dtype, autocast_dtype)
        if name not in ("bf16", "fp16"):
            raise ValueError(f"autocast_dtype must be 'bf16' / torch.bfloat16 or 'fp16' / torch.float16, got {autocast_dtype!r}")
        self.conv_precision = name
        st = {torch.bfloat16: "bf16", torch.float16: "fp16", torch.float32: "fp32", None: ("bf16" if name == "bf16" else "fp32")}.get(activation_storage, activation_storage)
        if st not in ("bf16", "fp16", "fp32"):
            raise ValueError(f"activation_storage must be 'bf16', 'fp16' or 'fp32', got {activation_storage!r}")
        if st != "fp32" and st != name:
            raise ValueError(f"activation_storage={st!r} goes with autocast_dtype={st!r}: a 16-bit tensor is stored in the type its convolutions "
                             "round their operands to (a plain input is the matrix operand as stored)")
        self.act_storage = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(st)


class HipAutoImplantUNet(HipUNet3D):
    """Drop-in for the reference's AutoImplantUNet (unet.py:61-70): forward returns y - x, `.test(x)` the plain network output
    (unet3d/predict/utils.py:46-47 prefers `.test` when present)."""

    def forward(self, x):
        return super().forward(x) - x

    def test(self, x):
        return super().forward(x)
