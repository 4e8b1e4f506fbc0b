"""Thin tensor-level wrappers over the C ABI (include/mi355_unet3d.h).

`Act` is an NDHWC fp32 view: a contiguous torch buffer [N, D, H, W, LD] plus a channel window [c0, c0+c).
A window narrower than LD is how torch.cat of the reference (segmentation/unet.py:42) is expressed: producers write
their channel slice of the concat buffer directly.

`Backend` owns a library handle, the stream getter and one growable workspace. The default backend is the HIP library
on the current CUDA(=HIP) device; it raises if the library is missing. (tests/ may construct a Backend around the CPU
emulator build of the same sources -- test infrastructure only, never used by the modules of this package.)
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (IN_AFFINE_ACT, IN_PLAIN, IN_S2D, IN_ZERO_INSERT, OUT_D2S, OUT_PLAIN, PREC_F32, PRECISIONS, W_OIDHW4,  # noqa: F401
                   W_PACKED, W_PACKED_F32_NARROW, MiAct, MiConvDesc, MiDiceOpts, MiGnBwdFuse, check)


# MI355_HOST_CACHES=0: the per-call forms of round 4 (descriptors, the Winograd routing query, the engine's parameter list and gradient
# views rebuilt every time) -- the A/B switch of the host-enqueue measurement (tools/host_enqueue.py, profiles/r5_host_enqueue.txt)
HOST_CACHES = os.environ.get("MI355_HOST_CACHES", "1") != "0"

ACT_DTYPES = {torch.float32: _lib.ACT_F32, torch.bfloat16: _lib.ACT_BF16, torch.float16: _lib.ACT_F16}      # storage types of an activation view


class Act:
    """`mom`: None, or the partial-moment records of this view's channels as a list of (records [N, B, c, 3], B, c) sources in
    channel order -- written by the epilogue of the conv that produced the tensor (Backend.conv_fwd(moments=True)) or by
    Backend.moments(); Backend.gn_stats finalises them instead of reading the tensor again.
    Storage type: the buffer's dtype, torch.float32, torch.bfloat16 or torch.float16 (mi355_act.dtype; HipAutocastUNet(activation_storage=...))."""
    __slots__ = ("buf", "c0", "c", "mom", "_d")

    def __init__(self, buf, c0=0, c=None):
        assert buf.dim() == 5 and buf.is_contiguous() and buf.dtype in ACT_DTYPES
        self.buf = buf
        self.c0 = c0
        self.c = buf.shape[-1] - c0 if c is None else c
        self.mom = None
        self._d = None
        assert self.c0 % 4 == 0 and self.c0 + self.c <= buf.shape[-1]

    @property
    def shape(self):  # logical (n, d, h, w, c)
        n, d, h, w, _ = self.buf.shape
        return (n, d, h, w, self.c)

    @property
    def ld(self):
        return self.buf.shape[-1]

    @property
    def dtype(self):
        return self.buf.dtype

    def ptr(self):
        return self.buf.data_ptr() + self.buf.element_size() * self.c0

    def desc(self):
        """The view's mi355_act, built once (buf / c0 / c are fixed at construction; the library only reads the struct)."""
        d_ = self._d
        if d_ is None or not HOST_CACHES:
            n, d, h, w, ld = self.buf.shape
            d_ = self._d = MiAct(self.ptr(), n, d, h, w, self.c, ld, ACT_DTYPES[self.buf.dtype])
        return d_

    def slice(self, c0, c):
        return Act(self.buf, self.c0 + c0, c)

    def tensor(self):
        """Logical view as a torch tensor [N, D, H, W, C] (strided)."""
        return self.buf[..., self.c0:self.c0 + self.c]

    def to_ncdhw(self):
        return self.tensor().permute(0, 4, 1, 2, 3).contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


_current_raw_stream = ((os.environ.get("MI355_RAW_STREAM", "1") != "0" and getattr(torch._C, "_cuda_getCurrentRawStream", None))
                       or (lambda i: torch.cuda.current_stream(i).cuda_stream))


class PackedWeight:
    """Kernel-layout copies of one conv weight, packed on first use per format: the fp32 MFMA layout and, when the backend
    runs its 3x3x3 stride-1 convolutions on the split-bf16 matrix path, the bf16-plane layout of that precision. Which one a
    conv call consumes is decided by the library (mi355_conv3d_uses_bf16)."""

    def __init__(self, be, w, mode):
        assert w.is_contiguous() and w.dtype == torch.float32
        self.be, self.w, self.mode = be, w, mode
        kd = w.shape[2]
        if mode in (0, 3):
            self.cout, self.cin = w.shape[0], w.shape[1]
        else:
            self.cout, self.cin = w.shape[1], w.shape[0]
        self.kd = kd
        self._f32 = None
        self._bf16 = {}

    def f32(self):
        if self._f32 is None:
            be = self.be
            n = be.lib.mi355_packed_weight_elems(self.cout, self.cin, self.kd, self.mode)
            out = torch.empty(n, dtype=torch.float32, device=self.w.device)
            check(be.lib.mi355_pack_conv_weight(self.w.data_ptr(), out.data_ptr(), self.cout, self.cin, self.kd, self.mode, be.stream()),
                  "pack_conv_weight")
            self._f32 = out
        return self._f32

    def bf16(self, precision):
        if precision not in self._bf16:
            be = self.be
            nb = be.lib.mi355_packed_weight_bytes_bf16(self.cout, self.cin, self.kd, precision)
            if nb == 0 or self.mode not in (0, 1):
                raise RuntimeError("no bf16 pack for this weight / precision")
            out = torch.empty((nb + 3) // 4, dtype=torch.int32, device=self.w.device)
            check(be.lib.mi355_pack_conv_weight_bf16(self.w.data_ptr(), out.data_ptr(), self.cout, self.cin, self.kd, self.mode, precision,
                                                     be.stream()), "pack_conv_weight_bf16")
            self._bf16[precision] = out
        self._bf16_last = precision
        return self._bf16[precision]

    def wino(self, form=None):
        """Winograd-domain weights of a 3x3x3 kernel, packed on first use: form "2d" = F(2x2, 3x3) x direct z (csrc/conv3d_wino.hip), "3d" =
        F(2x2x2, 3x3x3) (csrc/conv3d_wino3d.hip); None = the backend's current form."""
        form = form or self.be.wino_form
        if form == "auto":                                  # (resolved per call by Backend.conv_fwd; asked without a call: the 2-D pack)
            form = "2d"
        attr = "_wino3" if form == "3d" else "_wino"
        if getattr(self, attr, None) is None:
            setattr(self, attr, self.be.wino_pack_weight(self.w, self.mode, form))
        return getattr(self, attr)

    def ptr_for(self, desc):
        if (self.mode == 0 and self.cin == 4 and self.kd == 3 and desc.stride == 1 and desc.pad == 1 and desc.out_mode == OUT_PLAIN
                and desc.in_mode in (IN_PLAIN, IN_AFFINE_ACT)):
            desc.wformat = W_OIDHW4        # first layer: dedicated (tap, ci)-fused kernel reads the unpacked weight
            return self.w.data_ptr()
        if (self.cout <= 4 and self.kd == 3 and desc.stride == 1 and desc.pad == 1 and desc.in_mode == IN_PLAIN
                and desc.out_mode == OUT_PLAIN and not (desc.bias or desc.residual or desc.out_chscale)
                and desc.off_z == desc.off_y == desc.off_x == 0):
            desc.wformat = W_PACKED_F32_NARROW   # <= 4 output channels (first-layer dgrad): exact-fp32 vector-ALU kernel, fp32 pack
            return self.f32().data_ptr()
        if self.be.lib.mi355_conv3d_uses_bf16(ctypes.byref(desc)):
            return self.bf16(desc.precision).data_ptr()
        return self.f32().data_ptr()


class Backend:
    # Routing threshold of the Winograd kernels (voxels per sample). The first, 4-wave form showed no gain on the 16^3 level (256 channels:
    # profiles/r2_winograd_prep_measurement.txt, r3_winograd_landing.txt); the 8-wave kernel and the interleaved ring weight gradient do
    # (profiles/r3_wino_forms.txt section 5: forward 0.231 -> 0.136 ms, weight gradient 0.338 -> 0.220 ms, step 57.5 -> 56.0 ms). Below
    # 16^3 a launch has fewer workgroups than the chip has CUs: the direct kernels' smaller tiles stay.
    WINO_MIN_VOXELS = 16 ** 3
    # executed / algorithmic multiplications of the Winograd kernels (bench.py reports both rates)
    WINO_EXECUTED = {"conv3d_wino2d": 12.0 / 27.0, "conv3d_wino3d": 8.0 / 27.0, "conv3d_wgrad_wino_ring (+reduce)": 16.0 / 36.0}

    def __init__(self, lib=None, device=None):
        self.lib = lib if lib is not None else _lib.load_library()
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("3dunetcnn_amd needs an MI355X (no HIP device visible); there is no CPU fallback")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self._dev_index = None
        if self.device.type == "cuda":
            self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._wino_ok = {}          # call signature -> mi355_conv3d_wino_supported's answer (conv_fwd)
        self._ws_by_stream = {}     # launch stream handle -> workspace tensor: kernels of different streams must not share scratch
        self.precision = PREC_F32   # arithmetic of the 3x3x3 stride-1 convs: see set_precision()
        self.act_dtype = torch.float32   # storage type of the activations empty_act() makes (engine.HipNetBase sets it per network)
        # The eligible fp32 3x3x3 stride-1 forward / dgrad convolutions (>= WINO_MIN_VOXELS voxels, >= 8 channels either side) run on the
        # Winograd F(2x2, 3x3) x direct-z kernel (csrc/conv3d_wino.hip): 12 instead of 27 multiplications per output and (ci, co), fp32
        # error equal to the direct kernel's. Measured on MI355X (round 3, profiles/r3_winograd_landing.txt): layer set 21.96 -> 15.28 ms,
        # UNet3D 128^3 batch-2 step 87.1 -> 75.2 ms. MI355_WINOGRAD=0 selects the direct kernels (the A/B and cross-check form).
        self.winograd = os.environ.get("MI355_WINOGRAD", "1") == "1"
        # ... and which Winograd kernel: "3d" = F(2x2x2, 3x3x3) (csrc/conv3d_wino3d.hip, round 6: 8 multiplications per output and (ci, co)),
        # "2d" = F(2x2, 3x3) x direct z (csrc/conv3d_wino.hip: 12). MI355_WINO_FORM selects; the A/B is profiles/r6_wino3d.txt.
        # "auto" = per call: "3d" where it measured faster (profiles/r6_wino3d.txt: >= 256 input channels with a plain input, or at most
        # 16^3 voxels), "2d" elsewhere.
        self.wino_form = os.environ.get("MI355_WINO_FORM", "auto")
        if self.wino_form not in ("2d", "3d", "auto"):
            raise ValueError(f"MI355_WINO_FORM={self.wino_form!r}: '2d', '3d' or 'auto'")
        # Weight gradients of the same layers: "wino" = the plane-ring Winograd kernel (csrc/conv3d_wgrad_wino.hip: all three dz per
        # workgroup, every plane transformed once), "direct" = conv3d_wgrad_ring. Measured on MI355X (round 3,
        # profiles/r3_wgrad_wino_ring_ab.txt): 32->32 @128^3 1.93 -> 1.21 (-> 1.13) ms, layer set 1.55-1.7x, UNet3D step 74.7 -> 64.1 ms.
        # (A first Winograd weight gradient -- one dz per workgroup, planes transformed three times -- measured 2x SLOWER than the ring
        # kernel, profiles/r3_winograd_landing.txt, and was deleted.)
        self.wgrad_form = os.environ.get("MI355_WGRAD_FORM", "wino")
        if os.environ.get("MI355_WINO_MIN_VOXELS"):          # A/B of the routing threshold (forward / dgrad and weight gradient alike)
            self.WINO_MIN_VOXELS = int(os.environ["MI355_WINO_MIN_VOXELS"])

        # set to a list to collect one record per conv launch: (kernel family, flops, unfused-compulsory bytes [SURVEY 8d: inputs + outputs +
        # weights once], start event, end event, instantiation [the name a rocprofv3 trace shows, or None], bytes of the reads the
        # launch FUSES on top of the compulsory ones [residual, normalised tensor of the norm-backward sums]) -- see _prof_add
        self.prof = None
        # norm statistics leave with the producing conv's epilogue (csrc/gn_fuse.h). False: every statistic is a standalone pass
        # over the tensor again (the round-1 form; kept as the cross-check of the fused path, tests/test_ops_gpu.py)
        self.fused_stats = os.environ.get("MI355_FUSED_STATS", "1") != "0"

    def _prof_add(self, name, flops, byts, e0, e1, variant=None, fused_bytes=0.0):
        self.prof.append((name, flops, byts, e0, e1, variant, fused_bytes))

    def set_precision(self, name):
        """"fp32" (exact f32 MFMA, default) | "bf16x3" | "bf16x6" (split-bf16 fp32 emulation) | "bf16" | "fp16" (mixed precision: operands
        rounded to bf16 / IEEE fp16 while staged, fp32 accumulate; "fp16" is the arithmetic of torch's CUDA autocast). The tensors' storage
        type is a separate choice: `act_dtype` (fp32, or bf16 with the "bf16" mode -- HipAutocastUNet(activation_storage=...))."""
        self.precision = PRECISIONS[name] if isinstance(name, str) else int(name)

    # -- plumbing ------------------------------------------------------------------------------------------------
    def stream(self):
        """Raw handle of the device's CURRENT stream (what torch.cuda.current_stream(device).cuda_stream returns, without building a
        Stream object per launch: ~600 launches per step ask)."""
        if self._dev_index is not None:
            return _current_raw_stream(self._dev_index)
        return 0

    def ws(self, nbytes):
        """Workspace of the CURRENT launch stream (grown on demand, allocated under that stream so that the caching allocator
        ties its reuse to it). One per stream: the weight-gradient side stream of engine.py runs concurrently with the main
        stream, and a shared scratch buffer would be written by two kernels at once."""
        key = self.stream()
        cur = self._ws_by_stream.get(key)
        if cur is None or cur.numel() * 4 < nbytes:
            self._ws_by_stream[key] = None
            cur = self._ws_by_stream[key] = torch.empty((max(nbytes, 1 << 20) + 3) // 4, dtype=torch.float32, device=self.device)
        return cur


    def empty_act(self, n, d, h, w, c, ld=None, dtype=None):
        """dtype None: the backend's activation storage type (self.act_dtype: fp32, or bf16 inside a network that stores 16-bit activations)."""
        return Act(torch.empty(n, d, h, w, ld or c, dtype=dtype or self.act_dtype, device=self.device), 0, c)

    def zeros_act(self, n, d, h, w, c, ld=None, dtype=None):
        return Act(torch.zeros(n, d, h, w, ld or c, dtype=dtype or self.act_dtype, device=self.device), 0, c)

    def cast(self, x, dtype=None, out=None):
        """x in another storage type (mi355_cast): a new Act of `dtype`, or into `out`."""
        if out is None:
            out = self.empty_act(*x.shape, dtype=dtype)
        xd, yd = x.desc(), out.desc()
        check(self.lib.mi355_cast(ctypes.byref(xd), ctypes.byref(yd), self.stream()), "cast")
        return out

    # -- weights -------------------------------------------------------------------------------------------------
    def pack_weight(self, w, mode):
        """w: Conv3d weight [O, I, k, k, k] (modes 0, 1) or ConvTranspose3d weight [I, O, k, k, k] (modes 2, 3)."""
        return PackedWeight(self, w, mode)

    def repack_batch(self, packed, cache=None, precision=None):
        """Refresh the fp32, Winograd and 16-bit packs that the PackedWeights in `packed` hold from their (updated) weight tensors in ONE
        launch (mi355_pack_weights_batch; round 4: the 16-bit packs too -- dropped and rebuilt on first use they were 50 launches of 6 us
        per bf16 step). The device task table is cached in `cache`
        (a dict owned by the CALLER -- one per network, engine.HipNetBase._pack_tables -- mapping the task tuple to its device table):
        in a training loop neither the weights (views of the flat parameter buffer) nor the pack buffers move, so a network finds its
        table again every step, and two networks sharing this backend (validation twin, EMA copy) never evict each other's. A table is
        never replaced or freed while its owner lives: a captured HIP graph has the table's ADDRESS baked in (graph.py also holds a
        reference of its own)."""
        import numpy as np
        # precision: the mode of the forward that is about to run (engine passes its network's conv_precision); None: the backend's
        active = self.precision if precision is None else (PRECISIONS[precision] if isinstance(precision, str) else int(precision))
        CHUNK = 1024                                       # MI355_PACK_CHUNK work items (fp32: elements; Winograd: (dz, ci, co) triples)
        tasks, chunks = [], 0
        for pw in packed:
            cinP, coutP = (pw.cin + 7) // 8 * 8, (pw.cout + 31) // 32 * 32
            if pw._f32 is not None:
                tasks.append((pw.w.data_ptr(), pw._f32.data_ptr(), pw.cout, pw.cin, pw.kd, pw.mode, 0, chunks))
                chunks += (pw.kd ** 3 * cinP * coutP + CHUNK - 1) // CHUNK
            if getattr(pw, "_wino", None) is not None:
                tasks.append((pw.w.data_ptr(), pw._wino.data_ptr(), pw.cout, pw.cin, pw.kd, pw.mode, 1, chunks))
                chunks += (3 * cinP * coutP + CHUNK - 1) // CHUNK
            if getattr(pw, "_wino3", None) is not None:     # MI355_PACK_WINO3: one work item per (ci, co), input channels padded to 4
                tasks.append((pw.w.data_ptr(), pw._wino3.data_ptr(), pw.cout, pw.cin, pw.kd, pw.mode, 2, chunks))
                chunks += ((pw.cin + 3) // 4 * 4 * coutP + CHUNK - 1) // CHUNK
            # only the 16-bit pack of the precision mode the NEXT forward runs in (this backend's current mode) is refreshed: packs of
            # other modes -- every 16-bit pack once the backend is back in fp32, the other network's after two networks alternated
            # modes on one PackedWeight -- are dropped and rebuilt on first use instead of being rewritten every step for nobody
            pw._bf16 = {k: v for k, v in pw._bf16.items() if k == active}
            for prec, buf in pw._bf16.items():             # kind = MI355_PACK_LP + precision
                tasks.append((pw.w.data_ptr(), buf.data_ptr(), pw.cout, pw.cin, pw.kd, pw.mode, 16 + int(prec), chunks))
                chunks += (pw.kd ** 3 * ((pw.cin + 15) // 16 * 16) * coutP + CHUNK - 1) // CHUNK
        if not tasks:
            return 0
        if cache is None:
            cache = self.__dict__.setdefault("_pack_tables", {})     # callers without a cache of their own (tools, tests)
        key = tuple(tasks)
        table = cache.get(key)
        if table is None and self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # a new table needs a host-to-device copy, which a stream capture cannot contain: the single-weight launches do the same work
            for w, out, cout, cin, kd, mode, kind, _ in tasks:
                if kind == 0:
                    check(self.lib.mi355_pack_conv_weight(w, out, cout, cin, kd, mode, self.stream()), "pack_conv_weight")
                elif kind == 1:
                    check(self.lib.mi355_wino_pack_weight(w, out, cout, cin, mode, self.stream()), "wino_pack_weight")
                elif kind == 2:
                    check(self.lib.mi355_wino3d_pack_weight(w, out, cout, cin, mode, self.stream()), "wino3d_pack_weight")
                else:
                    check(self.lib.mi355_pack_conv_weight_bf16(w, out, cout, cin, kd, mode, kind - 16, self.stream()), "pack_conv_weight_bf16")
            self.last_pack_table = None                    # no table was used: nothing for a capture to pin
            return len(tasks)
        if table is None:
            rec = np.array(tasks, dtype=[("w", "<u8"), ("out", "<u8"), ("cout", "<i4"), ("cin", "<i4"), ("kd", "<i4"), ("mode", "<i4"),
                                         ("kind", "<i4"), ("first_chunk", "<i4")])
            assert rec.itemsize == 40                      # sizeof(mi355_pack_task)
            if len(cache) >= 8:                            # routing changed 8 times (precision / shape switches): drop the oldest unpinned
                for k in list(cache):
                    if not getattr(cache[k], "_mi355_pinned", 0):
                        del cache[k]
                        break
            table = cache[key] = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device)
        self.last_pack_table = table                       # the table THIS call used (engine.HipNetBase records it per network)
        check(self.lib.mi355_pack_weights_batch(table.data_ptr(), len(tasks), chunks, self.stream()), "pack_weights_batch")
        return len(tasks)

    # -- conv ----------------------------------------------------------------------------------------------------
    @staticmethod
    def _same_storage(what, t, ref, ref_name):
        """Operands handed to the library as RAW pointers (a conv's residual, the normalised tensor of the norm-backward sums, the addend
        of gn_act_bwd) carry no type on the C side: the header promises they have the storage type of y / dx, and nothing there can
        check it. A mismatch would be reinterpreted silently -- refuse it here."""
        if t is not None and t.dtype != ref.dtype:
            raise TypeError(f"{what} is stored as {t.dtype} but {ref_name} as {ref.dtype}: operands passed by pointer must share the "
                            "output's storage type (mi355_unet3d.h); cast first (Backend.cast)")

    def _desc(self, kd, stride, pad, in_mode, slope, scale, shift, bias, residual, chscale, off, out_dhw, keep, in_slope=None,
              out_mode=OUT_PLAIN):
        d = MiConvDesc()
        d.in_slope, d.out_mode, d.precision = _p(in_slope), out_mode, self.precision
        keep.append(in_slope)
        d.kd, d.stride, d.pad, d.in_mode, d.act_slope = kd, stride, pad, in_mode, slope
        d.in_scale, d.in_shift, d.bias = _p(scale), _p(shift), _p(bias)
        if residual is not None:
            d.residual, d.residual_ld = residual.ptr(), residual.ld
        d.out_chscale = _p(chscale)
        d.off_z, d.off_y, d.off_x = off
        d.out_d, d.out_h, d.out_w = out_dhw
        keep.extend([scale, shift, bias, residual, chscale])
        return d

    def conv_fwd(self, x, wp, y, kd, stride=1, pad=None, in_mode=IN_PLAIN, slope=0.0, scale=None, shift=None, bias=None,
                 residual=None, chscale=None, off=(0, 0, 0), out_dhw=None, in_slope=None, out_mode=OUT_PLAIN, moments=False, gnb=None):
        """moments=True: y will be normalised next -- have the epilogue emit its partial moments (y.mom) when this call can.
        gnb=(gx, stats, groups, slope): this call is a dgrad whose output is the gradient wrt act(norm(gx)) with stats =
        (mean_rstd, scale, shift) of gx: have the epilogue emit the first pass of gn_act_bwd. Returns None, or (records, B) to
        hand to gn_act_bwd(partials=...)."""
        pad = kd // 2 if pad is None else pad
        self._same_storage("conv_fwd: residual", residual, y, "y")
        if gnb is not None:
            self._same_storage("conv_fwd: normalised tensor of the norm-backward sums", gnb[0], y, "y")
        if out_dhw is None:
            out_dhw = x.shape[1:4] if out_mode == OUT_D2S else y.shape[1:4]
        if (self.winograd and self.precision == PREC_F32 and kd == 3 and stride == 1 and pad == 1 and in_mode in (IN_PLAIN, IN_AFFINE_ACT)
                and out_mode == OUT_PLAIN and tuple(off) == (0, 0, 0) and tuple(out_dhw) == tuple(y.shape[1:4]) and wp.mode in (0, 1)
                and wp.cin >= 8 and wp.cout >= 8 and x.shape[1:4] == y.shape[1:4]
                and x.shape[1] * x.shape[2] * x.shape[3] >= self.WINO_MIN_VOXELS):
            # the size thresholds are this layer's routing POLICY; whether the Winograd kernel can take the call at all (strides,
            # alignment of x / y / residual, modes) is the library's answer -- a call it refuses runs on the direct kernel below.
            # The answer is asked once per call signature (everything mi355_conv3d_wino_supported reads except the pointers' upper
            # bits); mi355_conv3d_wino_fwd checks the same conditions again on every call, so a stale entry fails loudly there
            form = self.wino_form
            if form == "auto":
                form = "3d" if x.c >= 256 and (in_mode == IN_PLAIN or x.shape[1] * x.shape[2] * x.shape[3] <= 16 ** 3) else "2d"
            key = (x.shape, x.ld, x.dtype, x.ptr() & 15, y.c, y.ld, y.dtype, in_mode, slope, scale is None, shift is None,
                   None if residual is None else residual.ld, form)
            ok = self._wino_ok.get(key) if HOST_CACHES else None
            if ok is None:
                probe = self._desc(3, 1, 1, in_mode, slope, scale, shift, bias, residual, chscale, (0, 0, 0), y.shape[1:4], [], in_slope, OUT_PLAIN)
                xd_, yd_ = x.desc(), y.desc()
                supported = self.lib.mi355_conv3d_wino3d_supported if form == "3d" else self.lib.mi355_conv3d_wino_supported
                ok = self._wino_ok[key] = bool(supported(ctypes.byref(xd_), ctypes.byref(yd_), ctypes.byref(probe)))
            if ok:
                return self.conv_fwd_wino(x, wp.wino(form), y, in_mode, slope, scale, shift, bias, residual, chscale, in_slope, moments, gnb)
        keep = []
        d = self._desc(kd, stride, pad, in_mode, slope, scale, shift, bias, residual, chscale, off, out_dhw, keep, in_slope, out_mode)
        xd, yd = x.desc(), y.desc()
        wptr = wp.ptr_for(d)                          # also settles d.wformat
        y.mom = None
        gparts = None
        if self.fused_stats and (moments or gnb is not None):
            if gnb is not None and not moments:
                # the query is for the norm-backward sums (a kernel may carry the moments epilogue but not this one): a placeholder
                # marks the request, the real descriptor replaces it below
                probe = MiGnBwdFuse()
                d.gn_bwd = ctypes.pointer(probe)
            nb = self.lib.mi355_conv3d_stats_blocks(ctypes.byref(xd), ctypes.byref(yd), ctypes.byref(d))
            d.gn_bwd = None
            if nb > 0 and moments:
                rec = torch.empty(x.shape[0], nb, y.c, 3, dtype=torch.float32, device=self.device)
                d.moments_out = rec.data_ptr()
                y.mom = [(rec, nb, y.c)]                      # folded after the launch (see below)
            elif nb > 0 and in_mode == IN_PLAIN and d.wformat == W_PACKED:
                gx, st, groups, gslope = gnb
                assert gx.shape == y.shape
                rec = torch.empty(x.shape[0], nb, y.c, 2, dtype=torch.float32, device=self.device)
                fuse = MiGnBwdFuse(gx.ptr(), gx.ld, st[1].data_ptr(), st[2].data_ptr(), st[0].data_ptr(), groups, gslope, rec.data_ptr())
                d.gn_bwd = ctypes.pointer(fuse)
                keep.extend([fuse, gx, st])
                gparts = (rec, nb)
        if self.prof is None:
            check(self.lib.mi355_conv3d_fwd(ctypes.byref(xd), wptr, ctypes.byref(yd), ctypes.byref(d), self.stream()), "conv3d_fwd")
            return self._fold_after(y, gparts)
        # profiling: HIP events on the launch stream around this one kernel, keyed by the kernel's trace name
        name = ctypes.create_string_buffer(96)
        self.lib.mi355_conv3d_fwd_config(ctypes.byref(xd), ctypes.byref(yd), ctypes.byref(d), name, 96)
        variant = None
        if d.wformat == W_OIDHW4:
            name.value = b"conv3d_c4_fwd"
        elif d.wformat == W_PACKED and self.lib.mi355_conv3d_uses_bf16(ctypes.byref(d)):
            # one family, several kernels: the tile forms and (16-bit single-product modes, eligible shapes) the plane-ring form;
            # mi355_conv3d_fwd_config names the instantiation this call launches exactly as a rocprofv3 trace prints it
            variant = name.value.decode()
            if os.environ.get("MI355_PROF_SHAPES") == "1":      # developer aid: one row per layer shape (tools/r6_call.sh)
                variant += f" [{x.c}->{y.c} @{out_dhw[0]} x{x.shape[0]}]"
            name.value = b"conv3d_k3_bf16<...>"
        nvox = x.shape[0] * out_dhw[0] * out_dhw[1] * out_dhw[2]
        flops = 2.0 * nvox * x.c * y.c * kd ** 3 * (8 if (in_mode == IN_S2D or out_mode == OUT_D2S) else 1)
        if in_mode == IN_ZERO_INSERT:
            flops /= 8.0   # algorithmic work of a stride-2 transposed conv: 27/8 taps per output voxel on average
        xb, yb = x.buf.element_size(), y.buf.element_size()          # algorithmic bytes follow the storage types
        byts = xb * (x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] * x.c) + yb * nvox * y.c + 4.0 * kd ** 3 * x.c * y.c
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(self.lib.mi355_conv3d_fwd(ctypes.byref(xd), wptr, ctypes.byref(yd), ctypes.byref(d), self.stream()), "conv3d_fwd")
        e1.record()
        fused = float(yb) * nvox * y.c * ((residual is not None) + (gparts is not None))
        self._prof_add(name.value.decode(), flops, byts, e0, e1, variant, fused)
        return self._fold_after(y, gparts)

    # -- Winograd form of the 3x3x3 stride-1 conv (csrc/conv3d_wino.hip) -------------------------------------------------------------
    def wino_pack_weight(self, w, mode=0, form=None):
        """w OIDHW [cout, cin, 3, 3, 3] -> transformed weights for conv_fwd_wino (mode 0: forward; mode 1: dgrad, i.e. a conv from
        cout to cin channels). form "2d" / "3d" (None: self.wino_form): which kernel's layout; the returned tensor remembers it
        (`mi355_form`), conv_fwd_wino launches the kernel that reads it."""
        form = form or ("2d" if self.wino_form == "auto" else self.wino_form)
        cout, cin = (w.shape[0], w.shape[1]) if mode == 0 else (w.shape[1], w.shape[0])
        same = w.device.type == self.device.type and (w.device.type != "cuda" or
                                                     (w.device.index if w.device.index is not None else torch.cuda.current_device()) ==
                                                     (self.device.index if self.device.index is not None else torch.cuda.current_device()))
        if not same or w.dtype != torch.float32:
            raise ValueError(f"wino_pack_weight: weight on {w.device} ({w.dtype}), backend on {self.device}: fp32 on the backend device expected")
        elems, pack = ((self.lib.mi355_wino3d_weight_elems, self.lib.mi355_wino3d_pack_weight) if form == "3d" else
                       (self.lib.mi355_wino_weight_elems, self.lib.mi355_wino_pack_weight))
        up = torch.empty(elems(cout, cin), dtype=torch.float32, device=self.device)
        check(pack(w.contiguous().data_ptr(), up.data_ptr(), cout, cin, mode, self.stream()), "wino_pack_weight")
        up.mi355_form = form
        return up

    def conv_fwd_wino(self, x, up, y, in_mode=IN_PLAIN, slope=0.0, scale=None, shift=None, bias=None, residual=None, chscale=None,
                      in_slope=None, moments=False, gnb=None):
        """Same contract as conv_fwd for a 3x3x3 stride-1 conv (moments / gnb: the fused statistics of the epilogue)."""
        self._same_storage("conv_fwd_wino: residual", residual, y, "y")
        if gnb is not None:
            self._same_storage("conv_fwd_wino: normalised tensor of the norm-backward sums", gnb[0], y, "y")
        keep = []
        d = self._desc(3, 1, 1, in_mode, slope, scale, shift, bias, residual, chscale, (0, 0, 0), y.shape[1:4], keep, in_slope, OUT_PLAIN)
        xd, yd = x.desc(), y.desc()
        y.mom = None
        gparts = None
        if self.fused_stats and (moments or gnb is not None):
            nb = self.lib.mi355_conv3d_wino_stats_blocks(ctypes.byref(yd))
            if moments:
                rec = torch.empty(x.shape[0], nb, y.c, 3, dtype=torch.float32, device=self.device)
                d.moments_out = rec.data_ptr()
                y.mom = [(rec, nb, y.c)]
            elif in_mode == IN_PLAIN:
                gx, st, groups, gslope = gnb
                rec = torch.empty(x.shape[0], nb, y.c, 2, dtype=torch.float32, device=self.device)
                fuse = MiGnBwdFuse(gx.ptr(), gx.ld, st[1].data_ptr(), st[2].data_ptr(), st[0].data_ptr(), groups, gslope, rec.data_ptr())
                d.gn_bwd = ctypes.pointer(fuse)
                keep.extend([fuse, gx, st])
                gparts = (rec, nb)
        if self.prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        form3 = getattr(up, "mi355_form", "2d") == "3d"
        fwd = self.lib.mi355_conv3d_wino3d_fwd if form3 else self.lib.mi355_conv3d_wino_fwd
        check(fwd(ctypes.byref(xd), up.data_ptr(), ctypes.byref(yd), ctypes.byref(d), self.stream()), "conv3d_wino3d_fwd" if form3 else "conv3d_wino_fwd")
        if self.prof is not None:
            e1.record()
            nvox = y.shape[0] * y.shape[1] * y.shape[2] * y.shape[3]
            fuse = 1 if d.moments_out else (2 if gparts is not None else 0)
            kern = "conv3d_wino3d" if form3 else "conv3d_wino2d_d8"
            self._prof_add("conv3d_wino3d" if form3 else "conv3d_wino2d", 2.0 * nvox * x.c * y.c * 27, 4.0 * (nvox * (x.c + y.c) + 27 * x.c * y.c), e0, e1,
                           f"{kern}<{1 if in_mode == IN_AFFINE_ACT else 0}, {fuse}>",
                           4.0 * nvox * y.c * ((residual is not None) + (gparts is not None)))
        return self._fold_after(y, gparts)

    def _fold_after(self, y, gparts):
        if y.mom is not None:
            rec, nb, c = y.mom[0]
            rec, nb = self._fold_records(rec, nb, c, 3)
            y.mom = [(rec, nb, c)]
        if gparts is not None:
            gparts = self._fold_records(gparts[0], gparts[1], y.c, 2)
        return gparts

    def conv_wgrad(self, x, dy, dw, kd, stride=1, pad=None, in_mode=IN_PLAIN, slope=0.0, scale=None, shift=None, in_slope=None,
                   out_mode=OUT_PLAIN):
        """out_mode=OUT_D2S: dy is the fine output of a ConvTranspose3d(k2,s2); dw is [8*dy.c, x.c] (parity-major rows)."""
        pad = kd // 2 if pad is None else pad
        keep = []
        d = self._desc(kd, stride, pad, in_mode, slope, scale, shift, None, None, None, (0, 0, 0),
                       x.shape[1:4] if out_mode == OUT_D2S else dy.shape[1:4], keep, in_slope, out_mode)
        xd, dyd = x.desc(), dy.desc()
        if (self.wgrad_form == "wino" and self.precision == PREC_F32 and kd == 3 and stride == 1 and pad == 1 and in_mode in (IN_PLAIN, IN_AFFINE_ACT)
                and out_mode == OUT_PLAIN and x.c >= 8 and dy.c >= 8 and x.shape[1] * x.shape[2] * x.shape[3] >= self.WINO_MIN_VOXELS):
            nbytes = self.lib.mi355_conv3d_wgrad_wino_workspace(ctypes.byref(xd), ctypes.byref(dyd), ctypes.byref(d))
            if nbytes:
                ws = self.ws(nbytes)
                assert dw.is_contiguous()
                if self.prof is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                check(self.lib.mi355_conv3d_wgrad_wino(ctypes.byref(xd), ctypes.byref(dyd), dw.data_ptr(), ctypes.byref(d), ws.data_ptr(),
                                                       ws.numel() * 4, self.stream()), "conv3d_wgrad_wino")
                if self.prof is not None:
                    e1.record()
                    nvox = dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3]
                    self._prof_add("conv3d_wgrad_wino_ring (+reduce)", 2.0 * nvox * x.c * dy.c * 27,
                                   4.0 * (nvox * (x.c + dy.c) + 27 * x.c * dy.c), e0, e1)
                return
        nbytes = self.lib.mi355_conv3d_wgrad_workspace(ctypes.byref(xd), ctypes.byref(dyd), ctypes.byref(d))
        if nbytes == 0:
            raise RuntimeError("conv3d_wgrad: unsupported configuration")
        ws = self.ws(nbytes)
        assert dw.is_contiguous()
        if self.prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(self.lib.mi355_conv3d_wgrad(ctypes.byref(xd), ctypes.byref(dyd), dw.data_ptr(), ctypes.byref(d), ws.data_ptr(),
                                          ws.numel() * 4, self.stream()), "conv3d_wgrad")
        if self.prof is not None:
            e1.record()
            nvox = dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3]
            flops = 2.0 * nvox * x.c * dy.c * kd ** 3
            byts = (x.buf.element_size() * (x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] * x.c) + dy.buf.element_size() * nvox * dy.c +
                    4.0 * kd ** 3 * x.c * dy.c)
            bf = self.precision != PREC_F32 and kd == 3 and stride == 1 and pad == 1 and in_mode in (IN_PLAIN, IN_AFFINE_ACT) and out_mode == OUT_PLAIN
            c4 = x.c == 4 and kd == 3 and stride == 1 and pad == 1 and in_mode in (IN_PLAIN, IN_AFFINE_ACT) and out_mode == OUT_PLAIN
            s2 = (kd == 3 and stride == 2 and pad == 1 and x.c == 32 and dy.c == 32 and in_mode == IN_PLAIN and out_mode == OUT_PLAIN
                  and os.environ.get("MI355_S2_KERNEL", "1") != "0")
            tr = (bf and x.buf.element_size() == 2 and x.c % 32 == 0 and dy.c % 32 == 0 and x.shape[2] >= 8 and x.shape[3] >= 16
                  and os.environ.get("MI355_WGRAD_LP_TR", "1") != "0")      # (plan_wt in csrc/conv3d_wgrad_lp.hip decides; this is the label)
            k1 = (kd == 1 and stride == 1 and in_mode == IN_PLAIN and out_mode == OUT_PLAIN and x.buf.dtype == dy.buf.dtype
                  and (x.c // 32, dy.c // 32) in ((1, 2), (2, 1), (2, 4), (4, 2)) and x.c % 32 == 0 and dy.c % 32 == 0
                  and os.environ.get("MI355_WGRAD_K1_STREAM", "1") != "0")  # (plan_wk1 in csrc/conv3d_wgrad_lp.hip decides; this is the label)
            self._prof_add("conv3d_wgrad_k1_stream (+reduce)" if k1 else "conv3d_c4_wgrad (+reduce)" if c4 else "conv3d_s2c32_wgrad (+reduce)" if s2 else "conv3d_wgrad_lp_tr (+reduce)" if tr
                           else "conv3d_wgrad_k3_bf16<...> (+reduce)" if bf
                           else "conv3d_wgrad_ring (+reduce)" if (kd == 3 and stride == 1 and pad == 1 and out_mode == OUT_PLAIN)
                           else f"conv3d_wgrad_mfma<{kd}, {stride}> (+reduce)", flops, byts, e0, e1)

    # -- first-layer backward in one pass over dy (csrc/conv3d_c4_bwd.hip) -----------------------------------------------------------------
    def c4_bwd_supported(self, x, dy, in_mode=IN_AFFINE_ACT, slope=0.0, scale=None, shift=None, in_slope=None):
        """Can the fused first-layer backward take this pair (fp32 4-channel input, 32-channel dy of any storage type; its arithmetic is exact fp32 in every precision mode)? MI355_C4_BWD=0: never."""
        if os.environ.get("MI355_C4_BWD", "1") == "0" or x.c != 4 or dy.c != 32:
            return False
        d = self._desc(3, 1, 1, in_mode, slope, scale, shift, None, None, None, (0, 0, 0), dy.shape[1:4], [], in_slope, OUT_PLAIN)
        xd, dyd = x.desc(), dy.desc()
        return bool(self.lib.mi355_conv3d_c4_bwd_supported(ctypes.byref(xd), ctypes.byref(dyd), ctypes.byref(d)))

    def c4_bwd(self, x, dy, wp, dw, groups, gamma, mean_rstd, scale, shift, dgamma, dbeta, slope=0.0, in_slope=None):
        """Backward of [GroupNorm(4 channels) -> act -> Conv3d(4 -> 32, k3)] whose input needs no gradient: dw (OIDHW) and dgamma / dbeta in
        one pass over dy, no data-gradient tensor. wp: PackedWeight of the conv weight, mode 1 (its fp32 pack is read)."""
        keep = []
        d = self._desc(3, 1, 1, IN_AFFINE_ACT, slope, scale, shift, None, None, None, (0, 0, 0), dy.shape[1:4], keep, in_slope, OUT_PLAIN)
        xd, dyd = x.desc(), dy.desc()
        nb = self.lib.mi355_conv3d_c4_bwd_blocks(ctypes.byref(xd))
        nbytes = self.lib.mi355_conv3d_c4_bwd_workspace(ctypes.byref(xd), ctypes.byref(dyd), ctypes.byref(d))
        if nb <= 0 or nbytes == 0:
            raise RuntimeError("c4_bwd: unsupported configuration (ask c4_bwd_supported first)")
        rec = torch.empty(x.shape[0], nb, 4, 2, dtype=torch.float32, device=self.device)
        ws = self.ws(max(nbytes, self.lib.mi355_gn_workspace(ctypes.byref(xd))))
        assert dw.is_contiguous()
        if self.prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(self.lib.mi355_conv3d_c4_bwd(ctypes.byref(xd), ctypes.byref(dyd), wp.f32().data_ptr(), dw.data_ptr(), ctypes.byref(d),
                                           mean_rstd.data_ptr(), groups, rec.data_ptr(), ws.data_ptr(), ws.numel() * 4, self.stream()), "conv3d_c4_bwd")
        if self.prof is not None:
            e1.record()
            nvox = dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3]
            # the two passes it replaces (weight gradient + data gradient, SURVEY 8d: each x + dy + w once): 2 x the forward's flops and
            # algorithmic bytes -- the fused kernel MOVES half of those bytes (dy and x once), which is the point
            self._prof_add("conv3d_c4_bwd (+reduce)", 2 * 2.0 * nvox * 4 * 32 * 27, 2 * 4.0 * (nvox * (4 + 32) + 27 * 4 * 32), e0, e1)
        # the records are few (<= ~1024 per sample): finalised in place of the norm backward's first pass; dgamma / dbeta only
        check(self.lib.mi355_gn_bwd_params(ctypes.byref(xd), groups, _p(gamma), mean_rstd.data_ptr(), _p(dgamma), _p(dbeta), rec.data_ptr(), nb,
                                           ws.data_ptr(), ws.numel() * 4, self.stream()), "gn_bwd_params")

    # -- norm ----------------------------------------------------------------------------------------------------
    RECORDS_MAX = 256        # more epilogue records than this per (sample, channel) are folded to RECORDS_FOLD before finalisation
    RECORDS_FOLD = 64

    def _fold_records(self, rec, nb, c, k):
        """(records, blocks) with at most RECORDS_MAX blocks: large layers leave thousands of per-tile records, which one workgroup
        per (sample, group) cannot walk quickly; one extra small launch folds them in parallel (mi355_gn_records_reduce)."""
        if nb <= self.RECORDS_MAX:
            return rec, nb
        out = torch.empty(rec.shape[0], self.RECORDS_FOLD, c, k, dtype=torch.float32, device=self.device)
        check(self.lib.mi355_gn_records_reduce(rec.data_ptr(), rec.shape[0], nb, c, k, out.data_ptr(), self.RECORDS_FOLD, self.stream()),
              "gn_records_reduce")
        return out, self.RECORDS_FOLD

    def moments(self, x):
        """Standalone producer of x.mom (one streaming read of x) for tensors no conv epilogue wrote: the trilinear-upsampled
        half of a concat buffer, the network input."""
        xd = x.desc()
        nb = self.lib.mi355_gn_moments_blocks(ctypes.byref(xd))
        rec = torch.empty(x.shape[0], nb, x.c, 3, dtype=torch.float32, device=self.device)
        check(self.lib.mi355_gn_moments(ctypes.byref(xd), rec.data_ptr(), self.stream()), "gn_moments")
        x.mom = [(rec, nb, x.c)]
        return x.mom

    def gn_stats(self, x, groups, eps, gamma, beta):
        """(mean_rstd [n, G, 2], scale [n, c], shift [n, c]) of GroupNorm(groups) over x: from the moment records the producer(s)
        of x left behind (x.mom, at most two sources: a concat buffer has two producers) or, without them, one standalone pass."""
        n, c = x.shape[0], x.c
        mean_rstd = torch.empty(n, groups, 2, dtype=torch.float32, device=self.device)
        scale = torch.empty(n, c, dtype=torch.float32, device=self.device)
        shift = torch.empty(n, c, dtype=torch.float32, device=self.device)
        mom = x.mom if self.fused_stats else None
        if mom is not None and 1 <= len(mom) <= 2 and sum(m[2] for m in mom) == c:
            a = mom[0]
            b = mom[1] if len(mom) == 2 else (None, 0, 0)
            check(self.lib.mi355_gn_finalize(a[0].data_ptr(), a[1], a[2], _p(b[0]), b[1], b[2], n, groups, eps, _p(gamma), _p(beta),
                                             mean_rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), self.stream()), "gn_finalize")
            return mean_rstd, scale, shift
        xd = x.desc()
        ws = self.ws(self.lib.mi355_gn_workspace(ctypes.byref(xd)))
        check(self.lib.mi355_gn_stats(ctypes.byref(xd), groups, eps, _p(gamma), _p(beta), mean_rstd.data_ptr(), scale.data_ptr(),
                                      shift.data_ptr(), ws.data_ptr(), ws.numel() * 4, self.stream()), "gn_stats")
        return mean_rstd, scale, shift

    def gn_act_bwd(self, x, dA, dx, groups, slope, gamma, mean_rstd, scale, shift, dgamma, dbeta, addend=None, partials=None):
        """partials: None, or what conv_fwd(gnb=...) returned for the dgrad that produced dA (its epilogue did the first pass)."""
        self._same_storage("gn_act_bwd: addend", addend, dx, "dx")
        xd, dad, dxd = x.desc(), dA.desc(), dx.desc()
        ws = self.ws(self.lib.mi355_gn_workspace(ctypes.byref(xd)))
        common = (ctypes.byref(xd), ctypes.byref(dad), ctypes.byref(dxd),
                  None if addend is None else addend.ptr(), 0 if addend is None else addend.ld,
                  groups, slope, _p(gamma), mean_rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), _p(dgamma), _p(dbeta))
        if partials is not None:
            check(self.lib.mi355_gn_act_bwd_fused(*common, partials[0].data_ptr(), partials[1], ws.data_ptr(), ws.numel() * 4, self.stream()),
                  "gn_act_bwd_fused")
        else:
            check(self.lib.mi355_gn_act_bwd(*common, ws.data_ptr(), ws.numel() * 4, self.stream()), "gn_act_bwd")

    # -- resample / layout / pointwise ---------------------------------------------------------------------------
    def upsample2x_fwd(self, lo, cat, off):
        a, b = lo.desc(), cat.desc()
        check(self.lib.mi355_upsample2x_fwd(ctypes.byref(a), ctypes.byref(b), off[0], off[1], off[2], self.stream()), "upsample2x_fwd")

    def upsample2x_bwd(self, dcat, dlo, off):
        a, b = dcat.desc(), dlo.desc()
        check(self.lib.mi355_upsample2x_bwd(ctypes.byref(a), ctypes.byref(b), off[0], off[1], off[2], self.stream()), "upsample2x_bwd")

    def ncdhw_to_ndhwc(self, src, dst):
        assert src.is_contiguous() and src.dtype == torch.float32
        d = dst.desc()
        check(self.lib.mi355_ncdhw_to_ndhwc(src.data_ptr(), ctypes.byref(d), self.stream()), "ncdhw_to_ndhwc")

    def ndhwc_to_ncdhw(self, src, dst):
        assert dst.is_contiguous() and dst.dtype == torch.float32
        s = src.desc()
        check(self.lib.mi355_ndhwc_to_ncdhw(ctypes.byref(s), dst.data_ptr(), self.stream()), "ndhwc_to_ncdhw")

    def add(self, a, b, y):
        ad, bd, yd = a.desc(), b.desc(), y.desc()
        check(self.lib.mi355_add(ctypes.byref(ad), ctypes.byref(bd), ctypes.byref(yd), self.stream()), "add")

    def chscale(self, x, s, y):
        xd, yd = x.desc(), y.desc()
        check(self.lib.mi355_chscale(ctypes.byref(xd), s.data_ptr(), ctypes.byref(yd), self.stream()), "chscale")

    # -- projection ----------------------------------------------------------------------------------------------
    def proj_fwd(self, x, w, bias, logits, scale=None, shift=None, slope=0.0):
        cout = w.shape[0]
        xd = x.desc()
        assert logits.is_contiguous()
        check(self.lib.mi355_proj_fwd(ctypes.byref(xd), _p(scale), _p(shift), slope, w.data_ptr(), _p(bias), logits.data_ptr(), cout,
                                      self.stream()), "proj_fwd")

    def proj_bwd(self, x, w, dlogits, dx, dw, dbias, scale=None, shift=None, slope=0.0):
        cout = w.shape[0]
        xd = x.desc()
        dxd = dx.desc() if dx is not None else None
        ws = self.ws(self.lib.mi355_proj_workspace(ctypes.byref(xd), cout))
        check(self.lib.mi355_proj_bwd(ctypes.byref(xd), _p(scale), _p(shift), slope, w.data_ptr(), dlogits.data_ptr(),
                                      ctypes.byref(dxd) if dxd is not None else None, dw.data_ptr(), _p(dbias), cout, ws.data_ptr(),
                                      ws.numel() * 4, self.stream()), "proj_bwd")

    # -- sliding-window inference ---------------------------------------------------------------------------------
    def sw_accumulate(self, pred, importance, out, count, start):
        """pred [C, rd, rh, rw], importance [rd, rh, rw], out [C, D, H, W], count [D, H, W] (fp32, contiguous)."""
        c, rd, rh, rw = pred.shape
        _, D, H, W = out.shape
        check(self.lib.mi355_sw_accumulate(pred.data_ptr(), importance.data_ptr(), out.data_ptr(), count.data_ptr(), c, rd, rh, rw,
                                           D, H, W, int(start[0]), int(start[1]), int(start[2]), self.stream()), "sw_accumulate")

    def sw_gather(self, volume, starts, roi):
        """volume [N, C, D, H, W]; starts: device int32 [nw, 4] = (sample, z0, y0, x0). Returns the window batch [nw, C, *roi]."""
        assert volume.is_contiguous() and volume.dtype == torch.float32 and starts.dtype == torch.int32 and starts.is_contiguous()
        n, c, D, H, W = volume.shape
        nw = starts.shape[0]
        win = torch.empty(nw, c, *roi, dtype=torch.float32, device=volume.device)
        check(self.lib.mi355_sw_gather(volume.data_ptr(), n, c, D, H, W, starts.data_ptr(), nw, roi[0], roi[1], roi[2], win.data_ptr(),
                                       self.stream()), "sw_gather")
        return win

    def sw_accumulate_batch(self, pred, importance, out, count, starts):
        """pred [nw, C, *roi], importance [*roi], out [N, C, D, H, W], count [N, D, H, W], starts device int32 [nw, 4]."""
        nw, c, rd, rh, rw = pred.shape
        n, _, D, H, W = out.shape
        check(self.lib.mi355_sw_accumulate_batch(pred.data_ptr(), importance.data_ptr(), out.data_ptr(), count.data_ptr(), n, c, rd, rh, rw,
                                                 D, H, W, starts.data_ptr(), nw, self.stream()), "sw_accumulate_batch")

    def sw_normalize(self, out, count):
        check(self.lib.mi355_sw_normalize(out.data_ptr(), count.data_ptr(), out.shape[0], count.numel(), self.stream()), "sw_normalize")

    # -- steps either side of the network ---------------------------------------------------------------------------
    def postprocess(self, logits, activation, threshold, labels, hierarchy, sum_then_threshold=False, want_probs=True, want_labels=True):
        """logits [C, D, H, W] fp32 (one sample). Returns (probs or None, int16 label map or None)."""
        assert logits.is_contiguous() and logits.dtype == torch.float32 and logits.dim() == 4
        c, vox = logits.shape[0], logits[0].numel()
        probs = torch.empty_like(logits) if want_probs else None
        lm = torch.empty(logits.shape[1:], dtype=torch.int16, device=logits.device) if want_labels else None
        lab = torch.as_tensor(list(labels), dtype=torch.int16, device=logits.device) if want_labels else None
        act = {None: 0, "none": 0, "sigmoid": 1, "softmax": 2}[activation]
        check(self.lib.mi355_postprocess(logits.data_ptr(), c, vox, act, float(threshold), _p(lab), int(bool(hierarchy)),
                                         int(bool(sum_then_threshold)), _p(probs), _p(lm), self.stream()), "postprocess")
        return probs, lm

    def one_hot(self, label_map, groups):
        """label_map [D, H, W] fp32; groups: list of lists of label values. Returns uint8 [len(groups), D, H, W]."""
        assert label_map.is_contiguous() and label_map.dtype == torch.float32
        vals = torch.tensor([float(v) for g in groups for v in g], dtype=torch.float32, device=label_map.device)
        offs, o = [0], 0
        for g in groups:
            o += len(g)
            offs.append(o)
        offs = torch.tensor(offs, dtype=torch.int32, device=label_map.device)
        out = torch.empty(len(groups), *label_map.shape, dtype=torch.uint8, device=label_map.device)
        check(self.lib.mi355_one_hot(label_map.data_ptr(), label_map.numel(), vals.data_ptr(), offs.data_ptr(), len(groups), out.data_ptr(),
                                     self.stream()), "one_hot")
        return out

    def zscore(self, x):
        """x [C, D, H, W] fp32 -> per-channel (x - mean) / std."""
        assert x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 4
        y = torch.empty_like(x)
        ws = self.ws(self.lib.mi355_zscore_workspace(x.shape[0]))
        check(self.lib.mi355_zscore(x.data_ptr(), y.data_ptr(), x.shape[0], x[0].numel(), ws.data_ptr(), ws.numel() * 4, self.stream()), "zscore")
        return y

    def resample_affine(self, src, out_shape, matrix, mode="trilinear", padding="border"):
        """src [C, D, H, W] fp32; matrix: 12 floats (3x4 row-major, dst voxel -> src voxel, (z, y, x) order). Returns [C, *out_shape]."""
        assert src.is_contiguous() and src.dtype == torch.float32 and src.dim() == 4
        m = (ctypes.c_float * 12)(*[float(v) for v in matrix])
        dst = torch.empty(src.shape[0], *out_shape, dtype=torch.float32, device=src.device)
        md = {"trilinear": 0, "bilinear": 0, "nearest": 1, "nearest_floor": 2}[mode]
        pd = {"border": 0, "zeros": 1}[padding]
        check(self.lib.mi355_resample_affine(src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], src.shape[2], src.shape[3],
                                             out_shape[0], out_shape[1], out_shape[2], m, md, pd, self.stream()), "resample_affine")
        return dst

    # -- loss / optimizer ----------------------------------------------------------------------------------------
    def dice(self, logits, target, sigmoid=True, batch=False, squared_pred=False, smooth_nr=1e-5, smooth_dr=1e-5,
             want_grad=True, grad_scale=1.0, generalized=False, include_background=True):
        assert logits.is_contiguous() and target.is_contiguous() and logits.dtype == torch.float32
        assert target.dtype in (torch.uint8, torch.float32) and target.shape == logits.shape
        n, c = logits.shape[0], logits.shape[1]
        vox = logits[0, 0].numel()
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        dlogits = torch.empty_like(logits) if want_grad else None
        ws = self.ws(self.lib.mi355_dice_workspace(n, c, vox))
        check(self.lib.mi355_dice_fwd_bwd(logits.data_ptr(), target.data_ptr(), 1 if target.dtype == torch.uint8 else 0, n, c, vox,
                                          int(sigmoid), int(batch), int(squared_pred), int(generalized), int(include_background), smooth_nr,
                                          smooth_dr, loss.data_ptr(),
                                          _p(dlogits), grad_scale, ws.data_ptr(), ws.numel() * 4, self.stream()), "dice_fwd_bwd")
        return loss, dlogits

    DICE_ACT = {None: 0, "sigmoid": 1, "softmax": 2}
    DICE_REDUCE = {"mean": 0, "sum": 1, "none": 2}

    def _dice_opts(self, target, activation, batch, squared_pred, include_background, jaccard, reduction, smooth_nr, smooth_dr, class_weight):
        kind = 2 if target.dtype == torch.int32 else (1 if target.dtype == torch.uint8 else 0)
        return MiDiceOpts(self.DICE_ACT[activation], kind, int(batch), int(squared_pred), int(include_background), int(jaccard),
                          self.DICE_REDUCE[reduction], float(smooth_nr), float(smooth_dr), _p(class_weight))

    def dice_ex_forward(self, logits, target, activation="sigmoid", batch=False, squared_pred=False, include_background=True, jaccard=False,
                        reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, class_weight=None):
        """monai DiceLoss with the options the fused mi355_dice_fwd_bwd does not carry. target: same shape as the logits (uint8 / fp32) or
        an int32 label map [n, 1, ...] (to_onehot_y). Returns (loss values [1] or one per term, state for dice_ex_backward)."""
        assert logits.is_contiguous() and target.is_contiguous() and logits.dtype == torch.float32
        n, c = logits.shape[0], logits.shape[1]
        vox = logits[0, 0].numel()
        assert (target.dtype == torch.int32 and target.numel() == n * vox) or (target.dtype in (torch.uint8, torch.float32) and target.shape == logits.shape)
        ce = c - (0 if include_background else 1)
        terms = (ce if batch else n * ce) if reduction == "none" else 1
        loss = torch.empty(terms, dtype=torch.float32, device=self.device)
        o = self._dice_opts(target, activation, batch, squared_pred, include_background, jaccard, reduction, smooth_nr, smooth_dr, class_weight)
        ws = torch.empty(self.lib.mi355_dice_workspace(n, c, vox) // 4, dtype=torch.float32, device=self.device)   # kept for backward
        check(self.lib.mi355_dice_ex_forward(ctypes.byref(o), logits.data_ptr(), target.data_ptr(), n, c, vox, loss.data_ptr(), ws.data_ptr(),
                                             ws.numel() * 4, self.stream()), "dice_ex_forward")
        return loss, (o, ws, class_weight)

    def dice_ex_backward(self, logits, target, state, upstream):
        """d(sum_t upstream[t] * loss[t]) / d(logits); upstream: fp32 tensor with one value per loss value."""
        o, ws, _keep = state
        n, c = logits.shape[0], logits.shape[1]
        vox = logits[0, 0].numel()
        dlogits = torch.empty_like(logits)
        upstream = upstream.reshape(-1).contiguous().float()
        check(self.lib.mi355_dice_ex_backward(ctypes.byref(o), logits.data_ptr(), target.data_ptr(), n, c, vox, upstream.data_ptr(),
                                              upstream.numel(), dlogits.data_ptr(), ws.data_ptr(), self.stream()), "dice_ex_backward")
        return dlogits

    def cross_entropy(self, logits, target, mode="softmax", weight=1.0, loss=None, dlogits=None, want_grad=True, grad_scale=1.0):
        """mode "softmax": CrossEntropyLoss(mean) with probability targets; "bce": BCEWithLogitsLoss(mean). `loss` / `dlogits`
        given: the weighted CE value / gradient is ADDED to them (fusing with a Dice term); else fresh tensors are returned."""
        assert logits.is_contiguous() and target.is_contiguous() and logits.dtype == torch.float32
        assert target.dtype in (torch.uint8, torch.float32) and target.shape == logits.shape
        n, c = logits.shape[0], logits.shape[1]
        vox = logits[0, 0].numel()
        acc_l, acc_g = loss is not None, dlogits is not None
        if loss is None:
            loss = torch.empty(1, dtype=torch.float32, device=self.device)
        if dlogits is None and want_grad:
            dlogits = torch.empty_like(logits)
        ws = self.ws(self.lib.mi355_ce_workspace(vox))
        check(self.lib.mi355_ce_fwd_bwd(logits.data_ptr(), target.data_ptr(), 1 if target.dtype == torch.uint8 else 0, n, c, vox,
                                        {"softmax": 0, "bce": 1}[mode], float(weight), loss.data_ptr(), int(acc_l), _p(dlogits), int(acc_g),
                                        float(grad_scale), ws.data_ptr(), ws.numel() * 4, self.stream()), "ce_fwd_bwd")
        return loss, dlogits

    def adam_step(self, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
        check(self.lib.mi355_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                                       weight_decay, step, grad_scale, self.stream()), "adam_step")


_default = {}


def default_backend(device=None):
    """HIP backend of `device` (an index, a torch.device or a tensor's .device; None: the current device). Created on first use;
    raises without library or GPU."""
    if not torch.cuda.is_available():
        raise RuntimeError("3dunetcnn_amd needs an MI355X (no HIP device visible); there is no CPU fallback")
    if isinstance(device, torch.device):
        device = device.index
    dev = torch.cuda.current_device() if device is None else int(device)
    if dev not in _default:
        _default[dev] = Backend(device=torch.device("cuda", dev))
        env = os.environ.get("MI355_PRECISION")        # fp32 (default) | bf16x6 | bf16x3 | bf16 | fp16
        if env:
            _default[dev].set_precision(env)
    return _default[dev]
