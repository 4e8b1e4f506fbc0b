"""Shared engine of the drop-in modules: flat parameter/gradient storage, packed-weight cache and the autograd bridge.

A network built on `HipNetBase` owns ordinary nn.Parameters (reference names/shapes, so state_dicts interchange) that are
views into ONE flat device buffer; its gradients are written by the HIP kernels into one flat gradient buffer (what the
fused Adam and the RCCL bucket all-reduce operate on). Subclasses implement `_forward_impl(x, keep)` and
`_backward_impl_body(be, saved, dlogits, need_dx)` with explicit forward/backward over the C ABI (no autograd graph inside).
"""
import contextlib
import os

import torch
import torch.nn as nn

from . import ops as _ops


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        need_grad = any(ctx.needs_input_grad)   # False under torch.no_grad() / for frozen inference
        try:
            logits, saved = model._forward_impl(x, need_grad)
        except BaseException:
            model._abort_forward()              # an exception inside the forward (an OOM the caller catches, a refused shape) must not
            raise                               # leave the shared backend in this network's precision / storage type
        ctx.model = model
        ctx.saved = saved
        ctx.x_requires_grad = x.requires_grad
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if ctx.saved is None:
            # the saved activations are released by the first backward (12 GB at 128^3, batch 2): there is no retain_graph form
            raise RuntimeError(f"{type(model).__name__}: backward called a second time through the same forward (or after a no-grad "
                               "forward): the saved activations are freed by the first backward; run the forward again")
        grads, dx = model._backward_impl(ctx.saved, dlogits.contiguous(), ctx.x_requires_grad)
        ctx.saved = None
        return (None, dx) + (None,) * len(grads)   # parameter .grad is set by _backward_impl (zero-copy views of the flat buffer)


# Module-tree epoch: bumped whenever ANY nn.Module in the process registers a parameter or a submodule (torch's global registration
# hooks: construction, `m.attr = Parameter(...)` / `= Module(...)`, parametrizations). HipNetBase._params() keeps its list of parameters
# while the epoch stands still instead of walking ~130 modules five times per training step (1.3 ms of host time per step).
_TREE_EPOCH = [0]


def _bump_tree_epoch(*_):
    _TREE_EPOCH[0] += 1


nn.modules.module.register_module_parameter_registration_hook(_bump_tree_epoch)
nn.modules.module.register_module_module_registration_hook(_bump_tree_epoch)


class HipNetBase(nn.Module):
    n_in_channels = None    # set by subclasses: expected input channel count

    def _init_engine(self):
        self._be = None
        self.conv_precision = None   # None: the backend's setting; "fp32" | "bf16x6" | "bf16x3" | "bf16" | "fp16": per-module override
        self.act_storage = None      # None: fp32 tensors; torch.bfloat16: activations and their gradients live in HBM as bf16 (ops.Act.dtype)
        self._flat = None          # flat parameter buffer (views are the nn.Parameters)
        self._flat_grad = None
        self._packed = {}          # id(param) -> [version, {mode: packed tensor}, data_ptr]
        self._packs_dirty = True
        self._packs_dirty_local = False
        self._pack_tables = {}     # task tuple -> device task table of Backend.repack_batch: per NETWORK, never evicted by another model
        self.grad_ready_callback = None        # set by ddp.GradientBucketReducer
        self.backward_start_callback = None
        self.grad_sync_callback = None         # set by ddp.GradientBucketReducer: joins the bucket all-reduces at the end of backward
        self.grad_accumulated_callback = None  # set by ddp.GradientBucketReducer: a backward that ACCUMULATED into existing .grad
        self._written = []
        # Weight-gradient kernels are enqueued on a second HIP stream so that the matrix-bound wgrads overlap the HBM-bound
        # norm-backward / upsample-backward passes of the dgrad chain (see _wgrad_stream). Measured on MI355X (round 2 A/B,
        # UNet3D 128^3 batch 2 fp32): 92.20 -> 91.01 ms per step, gradients bit-identical (tests/test_model_gpu.py).
        self.backward_side_stream = os.environ.get("MI355_SIDE_STREAM", "1") != "0"
        self._s2 = None
        self._s2_active = None

    # ---- flat parameter storage --------------------------------------------------------------------------------
    def _params(self):
        """list(self.parameters()), cached while the module tree is unchanged (see _TREE_EPOCH; conversions that replace Parameter
        objects -- .to_empty(), overwrite-on-conversion -- go through _apply, deletions through __delattr__)."""
        c = self.__dict__.get("_params_cache")
        if c is None or c[0] != _TREE_EPOCH[0] or not _ops.HOST_CACHES:
            c = self.__dict__["_params_cache"] = (_TREE_EPOCH[0], list(self.parameters()))
        return list(c[1])

    def invalidate_parameter_cache(self):
        """Drop the cached parameter list / gradient views: call after surgery on CHILD modules that torch's registration hooks do not see
        (`del net.encoder.layers[i]`, ModuleList.pop, `child.bias = None`, direct `_parameters[...]` writes). forward() also notices such
        changes by itself once per call (`_check_parameter_cache`)."""
        _bump_tree_epoch()

    def _check_parameter_cache(self):
        """Once per forward: one real walk of the module tree against the cached list (identity, order). torch's global hooks fire on
        registrations only; removals and replacements on child modules bump nothing, and a stale list would silently keep training the
        old parameters. 0.2 ms of host time per step against the 1.3 ms the cache saves."""
        c = self.__dict__.get("_params_cache")
        if c is None or c[0] != _TREE_EPOCH[0]:
            return
        cached = c[1]
        n = 0
        for p in self.parameters():
            if n >= len(cached) or cached[n] is not p:
                _bump_tree_epoch()
                return
            n += 1
        if n != len(cached):
            _bump_tree_epoch()

    def __getstate__(self):
        # the caches hold process-local ids and an epoch number: a pickled copy (torch.save(model)) must rebuild them
        d = dict(self.__dict__)
        d.pop("_params_cache", None)
        d.pop("_gview_cache", None)
        return d

    def _apply(self, fn, *args, **kwargs):
        _bump_tree_epoch()
        try:
            return super()._apply(fn, *args, **kwargs)
        finally:
            _bump_tree_epoch()

    def __delattr__(self, name):
        _bump_tree_epoch()
        super().__delattr__(name)

    def flatten_parameters(self):
        """(Re)point every nn.Parameter at a slice of one flat device buffer (16-byte aligned slices) so that the fused
        Adam and the gradient all-reduce run over single contiguous tensors. Idempotent; re-run after .cuda()/.to()."""
        ps = self._params()
        dev = ps[0].device
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        ok = self._flat is not None and self._flat.device == dev and self._flat.numel() == total and all(
            p.data_ptr() == self._flat.data_ptr() + 4 * o for p, o in zip(ps, offs))
        if ok:
            return self._flat
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(ps, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
        self._flat = flat
        self._offsets = offs
        self._flat_grad = None
        self.__dict__.pop("_gview_cache", None)
        self._packs_dirty = True
        return flat

    def flat_grad(self):
        if self._flat_grad is None or self._flat_grad.device != self._flat.device or self._flat_grad.numel() != self._flat.numel():
            self._flat_grad = torch.zeros_like(self._flat)
        return self._flat_grad

    def mark_parameters_updated(self):
        """Called by optimizers that write the flat buffer through raw pointers (no torch version bump)."""
        self._packs_dirty = True

    def _packed_weight(self, p, mode, transform=None):
        """Packed (kernel-layout) copy of weight `p`, cached until the parameter changes. `transform` (optional) maps the
        parameter tensor to the OIDHW tensor that is packed (used for ConvTranspose3d(k2,s2) -> 1x1x1 GEMM weights)."""
        ent = self._packed.get(id(p))
        if ent is None or ent[0] != p._version or self._packs_dirty_local or ent[2] != p.data_ptr():
            ent = [p._version, {}, p.data_ptr(), transform is not None]
            self._packed[id(p)] = ent
        if mode not in ent[1]:
            w = p.data if transform is None else transform(p.data).contiguous()
            ent[1][mode] = self._be.pack_weight(w, mode)      # lazily packed per format (ops.PackedWeight)
        return ent[1][mode]

    def _repack_stale(self, be, force):
        """Start of a forward after an optimizer step: every pack whose parameter changed (version counter, or `force` after
        mark_parameters_updated) is refreshed in ONE launch (Backend.repack_batch) instead of being dropped and rebuilt by ~70 single
        launches during the step. Entries whose weight moved, or whose pack source is a transformed copy (ConvTranspose3d(k2, s2) as a
        1x1x1 GEMM), are dropped and rebuilt lazily as before."""
        batch = []
        single = os.environ.get("MI355_BATCH_PACK", "1") == "0"      # A/B switch: the former one-launch-per-pack behaviour
        for p in self._params():
            ent = self._packed.get(id(p))
            if ent is None or not (force or ent[0] != p._version):
                continue
            if single or ent[3] or ent[2] != p.data_ptr():
                del self._packed[id(p)]
                continue
            ent[0] = p._version
            batch.extend(ent[1].values())
        if batch:
            be.repack_batch(batch, self._pack_tables, precision=self.conv_precision)      # the mode THIS network's forward runs in
            # the device task table this network's repack used (None: the per-weight launches ran). graph.HipGraphedTrainStep pins
            # exactly this one -- `be.last_pack_table` is shared by every model on the backend
            self._last_pack_table = be.last_pack_table

    # ---- forward / backward bridge -------------------------------------------------------------------------------
    def _check_input(self, x):
        if x.device.type != "cuda" and self._be is None:
            raise RuntimeError(f"{type(self).__name__} runs on an MI355X only: move the module and its input to the GPU "
                               "(.cuda()); there is no CPU fallback")
        if x.dim() != 5 or x.shape[1] != self.n_in_channels:
            raise ValueError(f"expected input [N, {self.n_in_channels}, D, H, W], got {tuple(x.shape)}")

    def _run(self, x):
        self._check_input(x)
        if x.shape[0] == 0:
            # empty batch (torch's Conv3d / GroupNorm return empty tensors for it): nothing to launch; the zero-size sum keeps
            # the autograd link, so loss.backward() yields zero-valued parameter gradients as it does for the reference
            n_out = getattr(self, "n_outputs", None) or getattr(self, "out_channels")
            return x.new_zeros((0, n_out) + tuple(x.shape[2:])) + sum(p.sum() * 0 for p in self._params())
        if getattr(self, "_is_replica", False):
            # torch.nn.DataParallel (what the reference wraps the model in for n_gpus > 1, unet3d/models/build.py:18-20) re-creates the
            # module per GPU every forward with its parameters as broadcast views: the flat parameter / gradient buffers, packed
            # weights and the explicit backward of this engine cannot live in such a replica
            raise RuntimeError(f"{type(self).__name__} does not run under torch.nn.DataParallel (n_gpus > 1 in the reference's "
                               "build_or_load_model): use one process per GPU -- `python bench.py --gpus N`, or "
                               "3dunetcnn_amd.ddp.GradientBucketReducer under torch.distributed (INTEGRATION.md, multi-GPU)")
        self._check_parameter_cache()
        flat = self.flatten_parameters()
        if self._be is not None and flat.device.type == "cuda" and self._be.device.type == "cuda" and flat.device != self._be.device:
            self._be = None                   # the module was moved (.to / .cuda(i)) since its last forward: take that device's backend
            self._packed = {}
        if x.device.type == "cuda" and flat.device.type == "cuda" and x.device != flat.device:
            raise RuntimeError(f"{type(self).__name__}: input on {x.device} but the module lives on {flat.device}")
        with self._device_guard(x):
            return _NetFunction.apply(self, x.contiguous().float(), *self._params())

    @staticmethod
    def _device_guard(x):
        """Make the input's device current for the forward (torch.empty / current_stream follow the current device)."""
        return torch.cuda.device(x.device) if x.device.type == "cuda" else contextlib.nullcontext()

    def _begin_forward(self):
        # the backend (library handle, workspace, launch stream) of the device the PARAMETERS live on -- not of whatever device
        # happens to be current: a model on cuda:1 must not allocate and launch on cuda:0
        if self._be is None:
            dev = self._flat.device if self._flat is not None else None
            self._be = _ops.default_backend(dev.index if dev is not None and dev.type == "cuda" else None)
        be = self._be
        if self._packed:
            self._repack_stale(be, self._packs_dirty)
            self._packs_dirty = False                      # entries dropped there are rebuilt on first use (their entry is gone)
        self._packs_dirty_local = self._packs_dirty
        self._saved_precision = be.precision
        self._saved_act_dtype = be.act_dtype
        if self.conv_precision is not None:
            be.set_precision(self.conv_precision)
        be.act_dtype = self.act_storage or torch.float32
        self._forward_open = True
        return be

    def _end_forward(self):
        self._packs_dirty = False
        self._packs_dirty_local = False
        self._be.precision = self._saved_precision
        self._be.act_dtype = self._saved_act_dtype
        self._forward_open = False

    def _abort_forward(self):
        """Undo _begin_forward's changes to the backend after a forward that raised (the packs stay marked dirty)."""
        if getattr(self, "_forward_open", False) and self._be is not None:
            self._be.precision = self._saved_precision
            self._be.act_dtype = self._saved_act_dtype
        self._forward_open = False

    def _grad_views(self, gbuf, ps):
        """id(parameter) -> its slice of the gradient buffer `gbuf`, shaped like the parameter. The views of the engine's own flat
        gradient buffer are made once and kept (two torch calls per parameter and backward otherwise: ~0.4 ms of host time per step);
        a scratch buffer of an accumulating backward gets fresh ones."""
        own = gbuf is self._flat_grad and _ops.HOST_CACHES
        c = self.__dict__.get("_gview_cache")
        if own and c is not None and c[0] is gbuf and c[1] == _TREE_EPOCH[0] and c[2] is self._offsets:
            return c[3]
        views = {id(p): gbuf[o:o + p.numel()].view(p.shape) for p, o in zip(ps, self._offsets)}
        if own:
            self.__dict__["_gview_cache"] = (gbuf, _TREE_EPOCH[0], self._offsets, views)
        return views

    def _gslice(self, p):
        self._written.append(p)
        return self._gviews[id(p)]

    @contextlib.contextmanager
    def _wgrad_stream(self, be, *used):
        """Context for a weight-gradient launch. Default: nothing (the launch goes to the current stream, in program order).
        With `backward_side_stream` the launch is forked onto the side stream after everything enqueued so far (its inputs are
        complete); `used` (Acts / tensors the kernel reads) are marked as in use by that stream so that the
        caching allocator does not hand their memory out again before the kernel has run. _backward_impl joins the streams."""
        s2 = self._s2_active
        if s2 is None:
            yield
            return
        ev = torch.cuda.Event()
        ev.record()
        s2.wait_event(ev)
        for t in used:
            if t is not None:
                (t.buf if hasattr(t, "buf") else t).record_stream(s2)
        with torch.cuda.stream(s2):           # its scratch: Backend.ws() keeps one workspace per launch stream
            yield

    def _flush_ready(self):
        """Report parameters whose gradient kernels have been enqueued (DDP launches the bucket all-reduce from here)."""
        if self._written and self.grad_ready_callback is not None:
            s2 = self._s2_active
            if s2 is None:
                self.grad_ready_callback(self._written)
            else:
                # side-stream form: the reported gradients were written partly on the current stream (norm gamma/beta) and partly on
                # the side stream (wgrads). Let the SIDE stream wait for the current one and launch the bucket all-reduces from it
                # (torch.distributed orders a collective after the stream that is current at the call), so the dgrad chain on the
                # current stream never waits for a weight gradient.
                ev = torch.cuda.Event()
                ev.record()
                s2.wait_event(ev)
                with torch.cuda.stream(s2):
                    self.grad_ready_callback(self._written)
        self._written = []

    def _backward_impl(self, saved, dlogits, need_dx):
        be = self._be
        ps = self._params()
        gbuf = self.flat_grad()
        accumulate = any(p.grad is not None for p in ps)
        reducer_cb = (self.grad_ready_callback, self.backward_start_callback, self.grad_sync_callback)
        if accumulate:
            # existing .grad tensors (which may alias the flat buffer) must be ADDED to: compute into scratch. With a DDP reducer attached
            # the per-bucket all-reduces cannot be launched from inside this backward (they would reduce the micro-batch, not the sum):
            # the reducer gets the accumulated buffer in one piece afterwards (grad_accumulated_callback; skipped under reducer.no_sync())
            gbuf = torch.zeros_like(self._flat)
            self.grad_ready_callback = self.backward_start_callback = self.grad_sync_callback = None
        self._gbuf = gbuf
        self._written = []
        if self.backward_start_callback is not None:
            self.backward_start_callback(gbuf)
        self._gviews = self._grad_views(gbuf, ps)
        self._packs_dirty_local = False
        saved_precision, saved_act_dtype = be.precision, be.act_dtype
        if self.conv_precision is not None:
            be.set_precision(self.conv_precision)
        be.act_dtype = self.act_storage or torch.float32
        self._s2_active = None
        if self.backward_side_stream and dlogits.device.type == "cuda":
            if self._s2 is None:
                self._s2 = torch.cuda.Stream(device=dlogits.device)
            self._s2_active = self._s2
        try:
            dx_t = self._backward_impl_body(be, saved, dlogits, need_dx)
            self._flush_ready()
        finally:
            be.precision = saved_precision
            be.act_dtype = saved_act_dtype
            if self._s2_active is not None:
                torch.cuda.current_stream().wait_stream(self._s2_active)     # join: the optimizer reads every gradient
                self._s2_active = None
            # also on an exception (an OOM the caller catches): a later backward must not run without its gradient exchange
            self.grad_ready_callback, self.backward_start_callback, self.grad_sync_callback = reducer_cb
        if self.grad_sync_callback is not None and not accumulate:
            # every bucket all-reduce has been launched by now: order the launch stream behind them (a stream-level wait on RCCL,
            # the host does not stall), so that ANY caller's `loss.backward(); optimizer.step()` -- the reference's epoch_training
            # loop, unet3d/train/training_utils.py:71-72 -- steps on fully reduced, averaged gradients without calling the reducer
            self.grad_sync_callback()
        grads = []
        gviews = self._gviews
        for p in ps:
            g = gviews[id(p)]
            grads.append(g)
            if not p.requires_grad:
                continue           # frozen parameter: as with autograd, no .grad appears (an optimizer holding it must not move it)
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        self._gbuf = self._gviews = None
        if accumulate and getattr(self, "grad_accumulated_callback", None) is not None:
            if all(p.grad is not None and p.grad.data_ptr() == self._flat_grad.data_ptr() + 4 * o for p, o in zip(ps, self._offsets) if p.requires_grad):
                self.grad_accumulated_callback(self._flat_grad)        # the sum of the micro-batch gradients, reduced in one piece
            else:
                raise RuntimeError("gradient accumulation with the DDP reducer needs the .grad tensors this engine created (views of its flat "
                                   "gradient buffer): do not replace p.grad between the micro-batches")
        return grads, dx_t
