"""Synthetic training batches: what bench.py, the tools and the tests feed the path with.

Image: randn fp32 [N, C, D, H, W] (a z-scored MRI stand-in, cf. NormalizeIntensityD(channel_wise=True) in
examples/brats2020/brats2020_config.json:140-144). Target: BraTS-like NESTED binary masks (hierarchy WT >= TC >= ET,
unet3d/scripts/script_utils.py:232-246): concentric ellipsoids around a jittered centre, stored as uint8 one-hot channels
(unet3d/transforms/one_hot.py:10). Pure torch on the host; no kernels involved.
"""
import torch


def synthetic_case(n, n_features, dhw, n_outputs=3, seed=0):
    """Synthetic inputs of SURVEY.md 8(d): randn image (z-scored MRI stand-in) + nested ellipsoid uint8 masks."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, n_features, *dhw, generator=g)
    d, h, w = dhw
    zz, yy, xx = torch.meshgrid(torch.arange(d), torch.arange(h), torch.arange(w), indexing="ij")
    y = torch.zeros(n, n_outputs, d, h, w, dtype=torch.uint8)
    fr = (0.30, 0.20, 0.10, 0.05, 0.03, 0.02, 0.01, 0.005)
    for i in range(n):
        c = [s / 2 + float(torch.rand(1, generator=g) * 2 - 1) * min(8.0, s / 8) for s in dhw]
        for k in range(n_outputs):
            f = fr[k]
            r = ((zz - c[0]) / (f * d)) ** 2 + ((yy - c[1]) / (f * h)) ** 2 + ((xx - c[2]) / (f * w)) ** 2
            y[i, k] = (r <= 1.0).to(torch.uint8)
    return x, y
