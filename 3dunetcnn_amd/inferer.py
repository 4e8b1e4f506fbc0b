"""HipSlidingWindowInferer: drop-in for monai.inferers.SlidingWindowInferer as the reference uses it.

The reference builds the inferer from config["inference"] (`getattr(monai.inferers, name)(**kwargs)`,
unet3d/scripts/script_utils.py:290-293) and calls it as `inferer(images, model)` in validation
(unet3d/train/training_utils.py:106-107) and prediction (unet3d/predict/volumetric.py:147-148). MONAI is un-vendored; the
semantics below follow MONAI >= 1.2 `sliding_window_inference` and are pinned by oracle/sliding_window_ref.py:

  * the volume is zero-padded (constant `cval`) up to roi_size where smaller, half of the deficit in front;
  * per axis scan interval = roi if roi == size else max(int(roi * (1 - overlap)), 1); window starts i * interval clipped to
    size - roi, up to the first window that reaches the end;
  * windows are run `sw_batch_size` at a time through the network;
  * output = sum_w importance * pred_w / sum_w importance with a constant or Gaussian (sigma = sigma_scale * roi, peak
    normalised to 1, floored at 1e-3) importance map; padding is cropped off.

MI355X shape: the window batch is gathered on the device, the network is the HIP path (no_grad forward keeps no activations),
and the weighted scatter-accumulate + final normalisation are HIP kernels (csrc/pointwise.hip) writing straight into the
full-size output -- no per-window host round trip.
"""
import math

import torch

from . import ops as _ops


def _scan_interval(image_size, roi_size, overlap):
    out = []
    for s, r, o in zip(image_size, roi_size, overlap):
        if r == s:
            out.append(int(r))
        else:
            iv = int(r * (1 - o))
            out.append(iv if iv > 0 else 1)
    return out


def _window_starts(image_size, roi_size, interval):
    per_dim = []
    for s, r, iv in zip(image_size, roi_size, interval):
        num = int(math.ceil(float(s) / iv))
        scan_dim = next((d for d in range(num) if d * iv + r >= s), None)
        n = scan_dim + 1 if scan_dim is not None else 1
        starts = []
        for i in range(n):
            st = i * iv
            st -= max(st + r - s, 0)
            starts.append(st)
        per_dim.append(starts)
    return [(z, y, x) for z in per_dim[0] for y in per_dim[1] for x in per_dim[2]]


def importance_map(roi_size, mode="constant", sigma_scale=0.125, device="cpu"):
    if mode == "constant":
        return torch.ones(tuple(roi_size), dtype=torch.float32, device=device)
    if mode != "gaussian":
        raise NotImplementedError(f"importance mode {mode!r} (MONAI: 'constant' | 'gaussian')")
    sig = sigma_scale if isinstance(sigma_scale, (list, tuple)) else [sigma_scale] * len(roi_size)
    w = None
    for r, s in zip(roi_size, sig):
        c = (r - 1) / 2.0
        g = torch.exp(-0.5 * ((torch.arange(r, dtype=torch.float32, device=device) - c) / (s * r)) ** 2)
        w = g if w is None else w[..., None] * g
    w = w / w.max()
    return torch.clamp(w, min=1e-3).contiguous()


class HipSlidingWindowInferer:
    def __init__(self, roi_size, sw_batch_size=1, overlap=0.25, mode="constant", sigma_scale=0.125, padding_mode="constant",
                 cval=0.0, sw_device=None, device=None, progress=False, cache_roi_weight_map=False, **unsupported):
        if unsupported:
            raise NotImplementedError("HipSlidingWindowInferer does not implement: " + ", ".join(sorted(unsupported)))
        if padding_mode != "constant":
            raise NotImplementedError("only padding_mode='constant' (MONAI default)")
        self.roi_size = tuple(roi_size) if isinstance(roi_size, (list, tuple)) else (roi_size,) * 3
        if len(self.roi_size) != 3:
            raise ValueError("roi_size must have 3 spatial entries")
        self.sw_batch_size = int(sw_batch_size)
        self.overlap = tuple(overlap) if isinstance(overlap, (list, tuple)) else (float(overlap),) * 3
        if any(o < 0 or o >= 1 for o in self.overlap):
            raise ValueError("overlap must be >= 0 and < 1")          # MONAI's message
        self.mode, self.sigma_scale, self.cval = str(mode).lower(), sigma_scale, float(cval)
        self._be = None
        self._w = None

    def __call__(self, inputs, network, *args, **kwargs):
        if inputs.device.type != "cuda" and self._be is None:
            raise RuntimeError("HipSlidingWindowInferer runs on an MI355X only (no CPU fallback)")
        be = self._be or _ops.default_backend(inputs.device)
        if inputs.dim() != 5:
            raise ValueError("expected inputs [N, C, D, H, W]")
        inputs = inputs.float()
        N = inputs.shape[0]
        image_size = list(inputs.shape[2:])
        roi = [r if r > 0 else s for r, s in zip(self.roi_size, image_size)]      # MONAI: non-positive roi entry = full extent
        pads = []
        for s, r in zip(image_size, roi):
            d = max(r - s, 0)
            pads.append((d // 2, d - d // 2))
        if any(p[0] or p[1] for p in pads):
            inputs = torch.nn.functional.pad(inputs, [pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]],
                                             mode="constant", value=self.cval)
        psize = list(inputs.shape[2:])
        starts = _window_starts(psize, roi, _scan_interval(psize, roi, self.overlap))
        if self._w is None or tuple(self._w.shape) != tuple(roi) or self._w.device != inputs.device:
            self._w = importance_map(roi, self.mode, self.sigma_scale, inputs.device)
        # MONAI's loop: the (sample, window) pairs of the whole batch in order, `sw_batch_size` of them per network call. Per call:
        # ONE gather launch builds the window batch from the volume, ONE accumulate launch folds its predictions (one thread per
        # output voxel walks the windows in order: deterministic, bitwise the per-window result).
        inputs = inputs.contiguous()
        plan = [(n, z, y, x) for n in range(N) for (z, y, x) in starts]
        plan_dev = torch.tensor(plan, dtype=torch.int32).to(inputs.device)       # one small H2D copy per volume batch
        out = count = None
        with torch.no_grad():
            for b0 in range(0, len(plan), self.sw_batch_size):
                st = plan_dev[b0:b0 + self.sw_batch_size].contiguous()
                win = be.sw_gather(inputs, st, roi)
                pred = network(win, *args, **kwargs)
                if isinstance(pred, (tuple, list)):
                    pred = pred[0]
                pred = pred.float().contiguous()
                if tuple(pred.shape[2:]) != tuple(roi):
                    raise NotImplementedError("networks that change the spatial size are not supported by HipSlidingWindowInferer")
                if out is None:
                    out = torch.zeros(N, pred.shape[1], *psize, dtype=torch.float32, device=inputs.device)
                    count = torch.zeros(N, *psize, dtype=torch.float32, device=inputs.device)
                be.sw_accumulate_batch(pred, self._w, out, count, st)
            for n in range(N):
                be.sw_normalize(out[n], count[n])
        res = out
        if any(p[0] or p[1] for p in pads):
            res = res[:, :, pads[0][0]:pads[0][0] + image_size[0], pads[1][0]:pads[1][0] + image_size[1],
                      pads[2][0]:pads[2][0] + image_size[2]].contiguous()
        return res
