"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce over RCCL/xGMI overlapped with backward.

Replaces the reference's single-process torch.nn.DataParallel (unet3d/models/build.py:18-20: per-step weight broadcast,
output gather and gradient reduce-add to GPU 0) with the MI355X-native shape: every rank owns a replica and its Adam
state; the only per-step exchange is the gradient all-reduce. GroupNorm/InstanceNorm and per-sample
Dice need no cross-rank statistics.

HipUNet3D writes parameter gradients into one flat buffer and reports each parameter as soon as its wgrad is enqueued
(`model.grad_ready_callback`). Buckets are contiguous ranges of that buffer (default ~25 MB: with xGMI's
point-to-point links a ring all-reduce is per-link bound at ~153 GB/s, so a 25 MB bucket costs ~0.3 ms while the
backward producing it takes several ms; the bucket holding the first encoder levels -- the last one backward completes,
hence the only exposed one -- is cut to ~3 MB); when the last parameter of a bucket is ready its all-reduce is launched with
async_op=True -- torch.distributed runs it on the communicator's own HIP stream after an event on the compute stream --
and `wait()` makes the compute stream wait for all of them. The engine calls `wait()` itself at the end of the explicit backward
(`model.grad_sync_callback`), so the reference's unmodified loop (`loss.backward(); optimizer.step()`,
unet3d/train/training_utils.py:71-72) steps on reduced gradients; calling it again is a no-op.

Averaging happens exactly once, here (ReduceOp.AVG on RCCL, SUM followed by 1/world on other backends): leave
`HipAdam.grad_scale` at 1.0 when a reducer is attached.
"""
import contextlib
import warnings

import torch.distributed as dist


class GradientBucketReducer:
    def __init__(self, model, bucket_bytes=25 << 20, process_group=None, average=True, reduce_single_rank=False):
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        self.bucket_bytes = bucket_bytes
        self.reduce_single_rank = reduce_single_rank     # issue the collectives even with one rank (exercises the RCCL path)
        self._works = []
        self._built_for = None
        self._unsynced_pending = False      # a no_sync() micro-batch has accumulated and not been reduced yet
        self._warned_accumulate = False
        model.grad_ready_callback = self._on_ready
        model.backward_start_callback = self._on_backward_start
        model.grad_sync_callback = self.wait        # the engine joins the exchange at the end of every backward
        model.grad_accumulated_callback = self._on_accumulated
        self._sync = True

    # -- setup ---------------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src=0):
        """One-time broadcast of rank `src`'s weights (replaces DataParallel's per-forward replicate)."""
        flat = self.model.flatten_parameters()
        if self.world > 1:
            dist.broadcast(flat, src=src, group=self.pg)
            self.model.mark_parameters_updated()

    def _build(self):
        m = self.model
        ps = list(m.parameters())
        offs = m._offsets
        total = m._flat.numel()
        # bucket boundaries over the flat layout
        # The flat layout follows parameter order (encoder first) while backward produces gradients in reverse, so the bucket at
        # the lowest offsets completes last and its all-reduce is the only one nothing overlaps with: keep that one small
        # (1/8 of a regular bucket: the first encoder levels), the rest regular.
        per = max(1, self.bucket_bytes // 4)
        bounds, start = [], 0
        while start < total:
            end = min(total, start + (max(1, per // 8) if start == 0 else per))
            bounds.append([start, end])
            start = end
        # snap boundaries to parameter starts so that no parameter straddles two buckets
        starts = sorted(offs)
        snapped = [0]
        for b in bounds[:-1]:
            cand = min(starts, key=lambda s: abs(s - b[1]))
            if cand > snapped[-1]:
                snapped.append(cand)
        snapped.append(total)
        self.buckets = [(snapped[i], snapped[i + 1]) for i in range(len(snapped) - 1)]
        self.bucket_of = {}
        self.pending_init = [0] * len(self.buckets)
        for p, o in zip(ps, offs):
            for bi, (a, b) in enumerate(self.buckets):
                if a <= o < b:
                    self.bucket_of[id(p)] = bi
                    self.pending_init[bi] += 1
                    break
        self._built_for = m._flat.data_ptr()

    # -- gradient accumulation ---------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def no_sync(self):
        """Micro-batches whose gradients only accumulate locally (torch DDP's no_sync): inside this context a backward launches no
        all-reduce. The first backward OUTSIDE the context that accumulates onto them reduces the accumulated buffer (averaged over
        ranks), so optimizer.step() sees mean-over-ranks of sum-over-micro-batches, as with torch.nn.parallel.DistributedDataParallel."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _on_accumulated(self, flat_grad):
        """A backward has ADDED its gradients onto existing ones (engine.py): reduce the accumulated flat buffer in one piece -- the
        per-bucket overlap is not available here, the micro-batch backward could not know it was the last."""
        if not self._sync:
            self._unsynced_pending = True
            return
        if not self._unsynced_pending and not self._warned_accumulate:
            # not the closing micro-batch of a no_sync() accumulation: either a plain step whose .grad tensors were kept
            # (zero_grad(set_to_none=False)) or gradient accumulation WITHOUT no_sync() (every micro-batch synced: numerically right, but
            # each backward re-reduces the whole running sum). Both lose the bucketed overlap every time
            self._warned_accumulate = True
            warnings.warn("GradientBucketReducer: backward found existing .grad tensors outside a no_sync() accumulation: the accumulated "
                          "gradients are all-reduced in ONE blocking piece after this backward (no per-bucket overlap). For plain steps use "
                          "optimizer.zero_grad(set_to_none=True); for gradient accumulation wrap all but the last micro-batch in "
                          "reducer.no_sync() so that only the closing backward exchanges", stacklevel=3)
        self._unsynced_pending = False
        self.allreduce_flat(flat_grad)

    def allreduce_flat(self, flat_grad):
        """The whole flat gradient buffer reduced (averaged) in ONE collective on the current stream. Used by the accumulation path
        above and by HipGraphedTrainStep (graph.py): a replayed HIP graph cannot call back into Python per bucket, so the exchange is
        one all-reduce between the replay and the optimizer step (96 MB over xGMI: ~1 ms, not overlapped with backward)."""
        if self.world == 1 and not self.reduce_single_rank:
            return
        if self.average and dist.get_backend(self.pg) == "nccl":
            dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG, group=self.pg)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
            if self.average:
                flat_grad.mul_(1.0 / self.world)

    # -- HIP-graph replay (graph.py) --------------------------------------------------------------------------------
    @contextlib.contextmanager
    def detached(self):
        """The model without this reducer's callbacks (capture / replay of a HIP graph: no Python runs inside a replayed backward);
        re-attached on exit, also on an exception."""
        m = self.model
        saved = (m.grad_ready_callback, m.backward_start_callback, m.grad_sync_callback, m.grad_accumulated_callback)
        m.grad_ready_callback = m.backward_start_callback = m.grad_sync_callback = m.grad_accumulated_callback = None
        try:
            yield
        finally:
            m.grad_ready_callback, m.backward_start_callback, m.grad_sync_callback, m.grad_accumulated_callback = saved

    # -- per-backward ----------------------------------------------------------------------------------------------
    def _on_backward_start(self, gbuf):
        if self._built_for != self.model._flat.data_ptr():
            self._build()
        self.gbuf = gbuf
        if not self._sync:
            self._unsynced_pending = True      # first micro-batch of a no_sync() accumulation
        self.pending = list(self.pending_init)
        self._works = []
        self.n_launched = 0            # bucket all-reduces launched during this backward (diagnostics / tests)

    def _on_ready(self, params):
        if not self._sync or (self.world == 1 and not self.reduce_single_rank):
            return
        for p in params:
            bi = self.bucket_of[id(p)]
            self.pending[bi] -= 1
            if self.pending[bi] == 0:
                a, b = self.buckets[bi]
                view = self.gbuf[a:b]
                self.n_launched += 1
                backend = dist.get_backend(self.pg)
                if self.average and backend == "nccl":
                    w = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
                    self._works.append((w, None))
                else:
                    w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    self._works.append((w, view if self.average else None))

    def wait(self):
        """Block the compute stream (not the host, on RCCL) until every bucket has been reduced."""
        for w, view in self._works:
            w.wait()
            if view is not None:
                view.mul_(1.0 / self.world)
        self._works = []
