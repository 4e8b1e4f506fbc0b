"""On-device versions of the steps either side of the network, same signatures as the reference's.

  compile_one_hot_encoding      unet3d/utils/one_hot.py:7-37   (LabelMapToOneHot, transforms/one_hot.py:7-16)
  convert_one_hot_to_label_map  unet3d/utils/one_hot.py:44-118 (threshold / argmax / hierarchy decode)
  activate_and_decode           unet3d/predict/volumetric.py:151-156 + the decode above, one pass over the logits
  normalize_intensity           MONAI NormalizeIntensityD(channel_wise=True, nonzero=False), datasets/segmentation.py:77-86

  resize                        MONAI ResizeD(spatial_size, mode=("trilinear", "nearest")) = F.interpolate(size=...,
                                align_corners=False), datasets/segmentation.py:63-68
  resample_to_match             MONAI ResampleToMatch(mode) of a prediction onto the source image grid,
                                predict/volumetric.py:135-136, 168-170 (voxel map inv(A_src) @ A_dst; parity unpinned: MONAI absent)

Inputs and outputs live on the GPU; there is no CPU fallback.
"""
import torch

from . import ops as _ops


def _be(t, be=None):
    if be is not None:
        return be
    if t.device.type != "cuda":
        raise RuntimeError("3dunetcnn_amd.prepost runs on an MI355X only (no CPU fallback)")
    return _ops.default_backend(t.device)


def compile_one_hot_encoding(data, n_labels, labels=None, dtype=torch.uint8, return_4d=True, round=True, _backend=None):
    be = _be(data, _backend)
    while data.dim() < 5:
        data = data[None]
    assert data.shape[1] == 1
    if not round:
        raise NotImplementedError("round=False")
    groups = []
    for i in range(n_labels):
        if labels is not None:
            groups.append(list(labels[i]) if isinstance(labels[i], (list, tuple)) else [labels[i]])
        else:
            groups.append([i + 1])
    outs = [be.one_hot(data[n, 0].float().contiguous(), groups) for n in range(data.shape[0])]
    y = torch.stack(outs).to(dtype)
    if return_4d:
        assert y.shape[0] == 1
        y = y[0]
    return y


def convert_one_hot_to_label_map(one_hot_encoding, labels, axis=0, threshold=0.5, sum_then_threshold=False, dtype=torch.int16,
                                 label_hierarchy=False, _backend=None):
    if axis != 0:
        raise NotImplementedError("axis != 0")
    be = _be(one_hot_encoding, _backend)
    if not label_hierarchy and all(isinstance(l, (list, tuple)) for l in labels):
        maps, i = [], 0
        for sub in labels:                                   # one label-map volume per label group (one_hot.py:51-61)
            maps.append(convert_one_hot_to_label_map(one_hot_encoding[i:i + len(sub)], sub, axis, threshold, sum_then_threshold, dtype,
                                                     False, be))
            i += len(sub)
        return torch.stack(maps, dim=0)
    x = one_hot_encoding[:len(labels)].float().contiguous()
    _, lm = be.postprocess(x, None, threshold, labels, label_hierarchy, sum_then_threshold, want_probs=False, want_labels=True)
    return lm.to(dtype)


def activate_and_decode(logits, activation, labels, threshold=0.5, label_hierarchy=False, sum_then_threshold=False, _backend=None):
    """logits [C, D, H, W] of one sample -> (probabilities [C, D, H, W], int16 label map [D, H, W]) in one kernel."""
    be = _be(logits, _backend)
    return be.postprocess(logits.float().contiguous(), activation, threshold, labels, label_hierarchy, sum_then_threshold)


def normalize_intensity(image, channel_wise=True, nonzero=False, _backend=None):
    """image [C, D, H, W] (one sample) or [N, C, D, H, W]."""
    if nonzero or not channel_wise:
        raise NotImplementedError("only channel_wise=True, nonzero=False (the shipped configs: brats2020_config.json:140-144)")
    be = _be(image, _backend)
    if image.dim() == 5:
        return torch.stack([be.zscore(image[n].float().contiguous()) for n in range(image.shape[0])])
    return be.zscore(image.float().contiguous())


def resize(img, spatial_size, mode="trilinear", _backend=None):
    """img [C, D, H, W] -> [C, *spatial_size]. "trilinear": F.interpolate(size=spatial_size, mode="trilinear",
    align_corners=False); "nearest": F.interpolate(mode="nearest") (source index floor(dst * in/out))."""
    be = _be(img, _backend)
    if mode not in ("trilinear", "nearest"):
        raise NotImplementedError(f"resize mode {mode!r}")
    x = img.float().contiguous()
    m = [0.0] * 12
    for i in range(3):
        sc = x.shape[1 + i] / float(spatial_size[i])
        m[4 * i + i] = sc
        m[4 * i + 3] = 0.5 * sc - 0.5 if mode == "trilinear" else 0.0
    return be.resample_affine(x, tuple(int(v) for v in spatial_size), m, "trilinear" if mode == "trilinear" else "nearest_floor", "border")


def resample_to_match(img, src_affine, dst_affine, dst_shape, mode="trilinear", padding_mode="border", _backend=None):
    """img [C, D, H, W] with voxel->world affine src_affine (4x4) resampled onto the grid (dst_shape, dst_affine):
    dst[c, v] = interp(img[c], inv(src_affine) @ dst_affine @ v). mode "trilinear"/"bilinear"/"nearest"; padding_mode
    "border"/"zeros"."""
    be = _be(img, _backend)
    a_src = torch.as_tensor(src_affine, dtype=torch.float64).cpu()
    a_dst = torch.as_tensor(dst_affine, dtype=torch.float64).cpu()
    m = (torch.linalg.inv(a_src) @ a_dst)[:3, :].reshape(-1).tolist()
    return be.resample_affine(img.float().contiguous(), tuple(int(v) for v in dst_shape), m, mode, padding_mode)
