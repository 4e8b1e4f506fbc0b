"""Input-side staging: pinned-memory, double-buffered host -> HBM transfer of the loader's batches on a copy
stream, with the on-device prologue (channel-wise z-score, label-map -> uint8 one-hot) applied before the batch is handed to
the training loop.

The reference iterates a torch DataLoader of {"image": Tensor, "label": Tensor} dicts and calls `.cuda()` on each batch inside
the step (unet3d/train/training_utils.py:40-42, 89-91), so the H2D copy of batch i+1 never overlaps the kernels of batch i and
the MONAI CPU pipeline does the intensity normalisation / one-hot encode (datasets/segmentation.py:53-86). `DeviceStager`
wraps the same loader object (it keeps `__len__`, which ProgressMeter needs, training_utils.py:29-32) and yields the same
dicts with device-resident tensors, so `.cuda()` in the loop is a no-op:

    loader = DeviceStager(loader, normalize=True, one_hot_labels=[[1, 2, 4], [1, 4], [4]])

There is no CPU fallback: without an MI355X the constructor raises.
"""
import torch

from . import ops as _ops


class DeviceStager:
    def __init__(self, loader, device=None, normalize=False, one_hot_labels=None, image_key="image", label_key="label", depth=2):
        if not torch.cuda.is_available():
            raise RuntimeError("3dunetcnn_amd.staging needs an MI355X (no HIP device visible); there is no CPU fallback")
        self.loader = loader
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.normalize, self.one_hot_labels = normalize, one_hot_labels
        self.image_key, self.label_key = image_key, label_key
        self.depth = max(1, int(depth))
        self.copy_stream = torch.cuda.Stream(self.device)
        # its own Backend: the prologue kernels run on the copy stream and must not share the compute stream's workspace
        self.be = _ops.Backend(device=self.device) if (normalize or one_hot_labels is not None) else None

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch):
        """Enqueue the H2D copies (and the device prologue) of one batch on the copy stream; returns (batch, ready event)."""
        out = {}
        with torch.cuda.stream(self.copy_stream):
            for k, v in batch.items():
                if not torch.is_tensor(v):
                    out[k] = v
                    continue
                if v.device.type == "cpu":
                    if not v.is_pinned():
                        v = v.contiguous().pin_memory()
                    out[k] = v.to(self.device, non_blocking=True)
                else:
                    out[k] = v
            if self.normalize and self.image_key in out:
                x = out[self.image_key].float().contiguous()
                out[self.image_key] = torch.stack([self.be.zscore(x[n]) for n in range(x.shape[0])])
            if self.one_hot_labels is not None and self.label_key in out:
                groups = [list(g) if isinstance(g, (list, tuple)) else [g] for g in self.one_hot_labels]
                lm = out[self.label_key].float().contiguous()
                if lm.dim() == 5:
                    lm = lm[:, 0]
                out[self.label_key] = torch.stack([self.be.one_hot(lm[n].contiguous(), groups) for n in range(lm.shape[0])])
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        queue = []
        done = False
        while True:
            while not done and len(queue) < self.depth:
                try:
                    queue.append(self._upload(next(it)))
                except StopIteration:
                    done = True
            if not queue:
                return
            batch, ev = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                    # the consumer's stream waits; the host does not
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)          # allocated on the copy stream, consumed on the compute stream
            yield batch
