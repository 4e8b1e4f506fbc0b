"""HipGraphedTrainStep: the inner loop of the reference's training step replayed as ONE HIP graph.

The reference's step (unet3d/train/training_utils.py:59-72: zero_grad -> model(images) -> criterion -> backward -> optimizer.step)
enqueues ~600 kernels per iteration through this package (26 convolutions forward, their dgrad/wgrad pairs, 52 norm passes ...).
At the 128^3 training patch the GPU needs ~47 ms per volume for them and the ~20 ms of host-side enqueue work hide behind it; at
small patches (the 64^3 plumbing configuration, validation crops, deep levels of a sliding-window sweep) the step is
launch-bound: the GPU finishes before the host has enqueued the next kernels. Everything between the input copy and the
optimizer is shape-static, sync-free and allocates only from torch's caching allocator, so it is captured once
(torch.cuda.CUDAGraph = hipGraph on ROCm: hipStreamBeginCapture on the stream the C ABI launches on) and replayed with a single
hipGraphLaunch per step.

What is inside the graph: weight (re)packing, forward, loss value + d(loss)/d(logits), the explicit backward writing every
parameter gradient into the model's flat gradient buffer. What stays outside: the copy of the new batch into the static input
tensors and the optimizer step (one fused Adam launch; kept eager so lr schedulers -- ReduceLROnPlateau in the reference's
config -- keep working and the bias-correction step count stays a host integer).

Multi-rank (a GradientBucketReducer attached to the model, ddp.py): the eager step launches its bucket all-reduces from Python
callbacks inside backward, which a replayed graph cannot do. The graphed step therefore captures forward + loss + backward WITHOUT
the callbacks and exchanges the whole flat gradient buffer in one all-reduce between the replay and the optimizer step
(`reducer.allreduce_flat`): per step the host enqueues one hipGraphLaunch, one collective and one Adam kernel instead of ~600
launches -- the form for many ranks on one host at short steps (the bf16 / small-patch configurations), at the price of an exchange
that is not overlapped with backward (96 MB over xGMI, ~1 ms). `capture=False` runs the same step uncaptured (the host logic of this
path on any device: the world-size-2 gloo test uses it with the CPU emulator).
"""
import contextlib
import torch


class HipGraphedTrainStep:
    def __init__(self, model, criterion, optimizer, example_x, example_y, warmup=2, capture=True):
        # the reducer attached to the model, if any (its bound method is the engine's grad_sync_callback)
        sync = getattr(model, "grad_sync_callback", None)
        self.reducer = getattr(sync, "__self__", None)
        if sync is not None and not hasattr(self.reducer, "allreduce_flat"):
            raise RuntimeError("HipGraphedTrainStep: the model's gradient callbacks do not belong to a GradientBucketReducer")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.graph = None
        if not capture:
            model.flatten_parameters()
            return
        if example_x.device.type != "cuda":
            raise RuntimeError("HipGraphedTrainStep runs on an MI355X only (no CPU fallback)")
        with (self.reducer.detached() if self.reducer is not None else contextlib.nullcontext()):
            self._capture(model, criterion, optimizer, example_x, example_y, warmup)

    def _capture(self, model, criterion, optimizer, example_x, example_y, warmup):
        # One captured stream: the graph already removes the launch gaps the fork hides. Capturing the weight gradients' side stream
        # as a second graph branch was measured in round 4 and lost: 58.1 vs 55.9 ms per fp32 step (eager: 56.0; profiles/r4_ab_experiments.txt).
        model.backward_side_stream = False
        self.x = example_x.detach().clone()          # static inputs: every replay reads these addresses
        self.y = example_y.detach().clone()
        model.flatten_parameters()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            # warm-up off the default stream: sizes the backend workspace, the flat gradient buffer and the allocator pools.
            # No optimizer step here -- constructing the graph must not change the weights.
            for _ in range(max(2, warmup)):
                optimizer.zero_grad(set_to_none=True)
                criterion(model(self.x), self.y).backward()
                model.mark_parameters_updated()      # the next forward repacks in one launch: its device task table exists before capture
        torch.cuda.current_stream().wait_stream(side)
        optimizer.zero_grad(set_to_none=True)        # capture writes the gradients in place (no accumulate branch)
        model.mark_parameters_updated()              # the pack kernels are part of every replay
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            logits = model(self.x)
            loss = criterion(logits, self.y)
            loss.backward()
        # The captured kernels have the ADDRESSES of the backend workspaces baked in. Backend.ws() keeps one workspace per launch stream;
        # the capture stream's was allocated during capture (from the graph's private pool, where stream-ordered reuse replays
        # faithfully). Hold a reference to every workspace for the life of the graph: a later eager op that needs a larger one makes
        # Backend.ws() allocate anew and drop its own reference, and without ours the caching allocator (the reference loop calls
        # empty_cache() every iteration, unet3d/train/training_utils.py:66) could hand a block the graph still writes to another tensor.
        be = model._be
        self._ws_refs = None if be is None else [t for t in be._ws_by_stream.values() if t is not None]
        # the same holds for the device task table of the one-launch weight repack (Backend.repack_batch): its address is in the graph.
        # Keep it alive and mark it so the owner's cache never drops it while this graph exists.
        # (the table of THIS model's repack, recorded by engine._repack_stale -- the backend-global attribute may belong to another model)
        self._pack_table = getattr(model, "_last_pack_table", None)
        if self._pack_table is not None:
            self._pack_table._mi355_pinned = getattr(self._pack_table, "_mi355_pinned", 0) + 1      # a count: several graphs may share it
        # ... and for every packed-weight buffer the captured convolutions read: Backend.repack_batch drops the 16-bit packs of precision
        # modes other than the one the NEXT eager forward runs in (an eager forward of this model in another precision after the capture
        # would free a pack the graph still reads). Our references keep the memory; the graph's own pack kernels keep it current.
        self._pack_refs = []
        for ent in getattr(model, "_packed", {}).values():
            for pw in ent[1].values():
                self._pack_refs.extend(t for t in (pw._f32, getattr(pw, "_wino", None), getattr(pw, "_wino3", None)) if t is not None)
                self._pack_refs.extend(pw._bf16.values())
        self.logits = logits.detach()                # static outputs, refreshed by every replay
        self.loss = loss.detach()
        self._grads = [(p, p.grad) for p in model.parameters() if p.grad is not None]

    def close(self):
        """Release the graph and unpin the repack task table it held (the owner's 8-entry table cache may evict it again)."""
        t, self._pack_table = getattr(self, "_pack_table", None), None
        if t is not None:
            t._mi355_pinned = max(0, getattr(t, "_mi355_pinned", 1) - 1)
        self.graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _exchange(self):
        if self.reducer is not None:
            self.reducer.allreduce_flat(self.model._flat_grad)

    def __call__(self, x, y):
        """One training step on the batch (x, y) (shapes/dtypes of the example batch). Returns the loss (a static 0-dim device
        tensor: read it with .item() before the next call if the value is needed)."""
        if self.graph is None:                       # capture=False: the same step, uncaptured
            self.optimizer.zero_grad(set_to_none=True)
            with (self.reducer.detached() if self.reducer is not None else contextlib.nullcontext()):
                loss = self.criterion(self.model(x), y)
                loss.backward()
            self._exchange()
            self.optimizer.step()
            return loss.detach()
        if x.shape != self.x.shape or y.shape != self.y.shape or y.dtype != self.y.dtype:
            raise ValueError(f"HipGraphedTrainStep was captured for x {tuple(self.x.shape)}, y {tuple(self.y.shape)} {self.y.dtype}")
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        if y.data_ptr() != self.y.data_ptr():
            self.y.copy_(y, non_blocking=True)
        self.graph.replay()
        for p, g in self._grads:                     # a zero_grad(set_to_none=True) between calls must not hide the gradients
            p.grad = g
        self._exchange()                             # multi-rank: one all-reduce of the flat gradient buffer
        self.optimizer.step()
        return self.loss
