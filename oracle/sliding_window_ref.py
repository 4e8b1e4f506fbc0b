"""TEST INFRASTRUCTURE (oracle) -- plain-torch CPU restatement of MONAI `sliding_window_inference` /
`SlidingWindowInferer` as the reference uses it (inferer built at unet3d/scripts/script_utils.py:290-293, called at
unet3d/train/training_utils.py:106-107 and unet3d/predict/volumetric.py:147-148).

PARITY UNPINNED: MONAI is un-vendored (requirements.txt:4) and not installed here; the reference's tests hold no fixture for
it. Restates the published MONAI >= 1.2 algorithm (monai/inferers/utils.py: sliding_window_inference, _get_scan_interval;
monai/data/utils.py: dense_patch_slices, compute_importance_map) -- SURVEY.md 8f-1. The known-answer in SURVEY 8d (240x240x155,
roi 128, overlap 0.5 -> 3*3*2 = 18 windows with starts {0,64,112}^2 x {0,27}) is checked by tests/test_inferer.py.
"""
import math

import torch
import torch.nn.functional as F


def scan_interval(image_size, roi_size, overlap):
    out = []
    for s, r in zip(image_size, roi_size):
        if r == s:
            out.append(int(r))
        else:
            iv = int(r * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return out


def dense_patch_starts(image_size, patch_size, interval):
    starts = []
    for s, r, iv in zip(image_size, patch_size, interval):
        num = int(math.ceil(float(s) / iv))
        scan_dim = None
        for d in range(num):
            if d * iv + r >= s:
                scan_dim = d
                break
        n = scan_dim + 1 if scan_dim is not None else 1
        dim = []
        for i in range(n):
            st = i * iv
            st -= max(st + r - s, 0)
            dim.append(st)
        starts.append(dim)
    return [(z, y, x) for z in starts[0] for y in starts[1] for x in starts[2]]


def gaussian_importance(roi, sigma_scale=0.125):
    grids = []
    for r in roi:
        c = (r - 1) / 2.0
        grids.append(torch.exp(-0.5 * ((torch.arange(r, dtype=torch.float32) - c) / (sigma_scale * r)) ** 2))
    w = grids[0][:, None, None] * grids[1][None, :, None] * grids[2][None, None, :]
    w = w / w.max()
    return torch.clamp(w, min=1e-3)


def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25, mode="constant", sigma_scale=0.125, cval=0.0):
    image_size = list(inputs.shape[2:])
    roi = list(roi_size)
    pad = []
    for k in range(2, -1, -1):
        d = max(roi[k] - image_size[k], 0)
        pad.extend([d // 2, d - d // 2])
    x = F.pad(inputs, pad, mode="constant", value=cval)
    psize = list(x.shape[2:])
    starts = dense_patch_starts(psize, roi, scan_interval(psize, roi, overlap))
    w = torch.ones(roi) if mode == "constant" else gaussian_importance(roi, sigma_scale)
    out = cnt = None
    N = x.shape[0]
    total = [(n, s) for n in range(N) for s in starts]
    for b0 in range(0, len(total), sw_batch_size):
        chunk = total[b0:b0 + sw_batch_size]
        win = torch.stack([x[n, :, z:z + roi[0], y:y + roi[1], xx:xx + roi[2]] for n, (z, y, xx) in chunk])
        pred = predictor(win)
        if out is None:
            out = torch.zeros(N, pred.shape[1], *psize)
            cnt = torch.zeros(N, 1, *psize)
        for i, (n, (z, y, xx)) in enumerate(chunk):
            out[n, :, z:z + roi[0], y:y + roi[1], xx:xx + roi[2]] += w * pred[i]
            cnt[n, :, z:z + roi[0], y:y + roi[1], xx:xx + roi[2]] += w
    out = out / cnt
    zs, ys, xs = pad[4], pad[2], pad[0]
    return out[:, :, zs:zs + image_size[0], ys:ys + image_size[1], xs:xs + image_size[2]]
