"""TEST INFRASTRUCTURE (oracle) -- plain-torch CPU restatement of the reference's label encode/decode and of the intensity
normalisation it configures.

  compile_one_hot_encoding         follows unet3d/utils/one_hot.py:7-37 (+ isclose :40-43)
  convert_one_hot_to_label_map     follows unet3d/utils/one_hot.py:44-118
  normalize_intensity              MONAI NormalizeIntensity(channel_wise=True, nonzero=False) [un-vendored; published algorithm:
                                   per channel (x - mean) / std with torch.std(unbiased=False), divisor 0 -> 1]

PINNED for the first two: tests/test_prepost.py checks them (a) against the reference's own functions imported live from
/root/reference (build container), and (b) against the known answers of the reference's own tests
(test/test_utils.py:103-128 test_compile_one_hot_encoding, test/test_segment.py:8-19 test_segment_left_right), rebuilt from their
descriptions. normalize_intensity: parity unpinned (MONAI).
"""
import torch


def _isclose(a, b, atol=1e-8, rtol=1e-5):
    return torch.isclose(torch.ones(1) * a, torch.ones(1) * b, atol=atol, rtol=rtol)


def compile_one_hot_encoding(data, n_labels, labels=None, dtype=torch.uint8, return_4d=True):
    while data.dim() < 5:
        data = data[None]
    data = torch.round(data.float(), decimals=0)
    y = torch.zeros([data.shape[0], n_labels] + list(data.shape[2:]), dtype=dtype)
    for i in range(n_labels):
        if labels is not None:
            group = labels[i] if isinstance(labels[i], (list, tuple)) else [labels[i]]
        else:
            group = [i + 1]
        for lab in group:
            y[:, i][_isclose(data[:, 0], lab)] = 1
    return y[0] if return_4d else y


def convert_one_hot_to_label_map(one_hot, labels, threshold=0.5, sum_then_threshold=False, dtype=torch.int16, label_hierarchy=False):
    if label_hierarchy:
        roi = torch.ones(one_hot.shape[1:], dtype=torch.bool)
        lm = torch.zeros(roi.shape, dtype=dtype)
        for i, lab in enumerate(labels):
            roi = torch.logical_and(one_hot[i] > threshold, roi)
            lm[roi] = lab
        return lm
    if all(isinstance(l, (list, tuple)) for l in labels):
        maps, i = [], 0
        for sub in labels:
            maps.append(convert_one_hot_to_label_map(one_hot[i:i + len(sub)], sub, threshold, sum_then_threshold, dtype))
            i += len(sub)
        return torch.stack(maps, 0)
    n = len(labels)
    mask = (torch.sum(one_hot[:n], 0) > threshold) if sum_then_threshold else torch.any(one_hot[:n] > threshold, 0)
    arg = torch.zeros(one_hot.shape[1:], dtype=dtype)
    arg[mask] = (torch.argmax(one_hot[:n], 0) + 1)[mask].to(dtype)
    lm = torch.zeros_like(arg)
    for i, lab in enumerate(labels):
        lm[arg == (i + 1)] = lab
    return lm


def normalize_intensity(x):
    out = torch.empty_like(x, dtype=torch.float32)
    for c in range(x.shape[0]):
        m, s = x[c].float().mean(), x[c].float().std(unbiased=False)
        out[c] = (x[c].float() - m) / (s if float(s) != 0.0 else 1.0)
    return out


def resize_ref(img, spatial_size, mode="trilinear"):
    """MONAI Resize (datasets/segmentation.py:63-68) = torch F.interpolate on the channel-first sample; align_corners=False."""
    import torch.nn.functional as F
    if mode == "trilinear":
        return F.interpolate(img[None].float(), size=tuple(spatial_size), mode="trilinear", align_corners=False)[0]
    return F.interpolate(img[None].float(), size=tuple(spatial_size), mode="nearest")[0]


def resample_to_match_ref(img, src_affine, dst_affine, dst_shape, mode="trilinear", padding_mode="border"):
    """ResampleToMatch (predict/volumetric.py:135-136, 168-170) restated as torch F.grid_sample(align_corners=True) over the voxel
    map inv(A_src) @ A_dst (float64 coordinates -> normalised grid). PARITY UNPINNED: MONAI is not importable here; the voxel
    map is the affine contract of the two NIfTI grids, MONAI's own sub-voxel conventions are not checked."""
    import torch
    import torch.nn.functional as F
    a = torch.linalg.inv(torch.as_tensor(src_affine, dtype=torch.float64)) @ torch.as_tensor(dst_affine, dtype=torch.float64)
    dd, dh, dw = dst_shape
    zz, yy, xx = torch.meshgrid(torch.arange(dd, dtype=torch.float64), torch.arange(dh, dtype=torch.float64),
                                torch.arange(dw, dtype=torch.float64), indexing="ij")
    v = torch.stack([zz, yy, xx, torch.ones_like(zz)], dim=-1) @ a.T          # (..., 4): source (z, y, x, 1)
    sd, sh, sw = img.shape[1:]
    def norm(c, n):
        return 2.0 * c / max(n - 1, 1) - 1.0
    grid = torch.stack([norm(v[..., 2], sw), norm(v[..., 1], sh), norm(v[..., 0], sd)], dim=-1)[None]
    gm = {"trilinear": "bilinear", "bilinear": "bilinear", "nearest": "nearest"}[mode]
    return F.grid_sample(img[None].double(), grid, mode=gm, padding_mode=padding_mode, align_corners=True)[0].float()
