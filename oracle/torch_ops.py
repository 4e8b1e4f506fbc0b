"""TEST INFRASTRUCTURE (oracle) -- CPU fp32 restatement of each fused op of the hot path in plain torch.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this. Each function cites the
reference lines it follows; the arithmetic itself is torch ATen on CPU (the reference's own arithmetic provider,
SURVEY.md section 8c). Parity: see oracle/README.md -- the reference pins nothing for this path ("parity unpinned");
the whole-network oracle (oracle/reference_shim.py) runs the reference's own module graph.
"""
import torch
import torch.nn.functional as F


def norm_act(x, groups, gamma, beta, eps=1e-5, slope=0.0):
    """GroupNorm -> ReLU (myronenko.py:17-20) / InstanceNorm3d -> LeakyReLU(0.01) (DynUNet)."""
    u = F.group_norm(x, groups, gamma, beta, eps)
    return F.leaky_relu(u, slope) if slope != 0.0 else F.relu(u)


def conv_block(x, w, stride=1, pad=None, norm=None, bias=None, residual=None, chscale=None):
    """[GroupNorm -> ReLU ->] Conv3d(bias=False) [+ residual] [* dropout scale]
    (myronenko.py:17-21, 47-58, 75-80; resnet.py:12-22). norm = (groups, gamma, beta, eps, slope) or None."""
    if pad is None:
        pad = w.shape[2] // 2
    a = norm_act(x, *norm) if norm is not None else x
    y = F.conv3d(a, w, bias, stride=stride, padding=pad)
    if residual is not None:
        y = y + residual
    if chscale is not None:
        y = y * chscale[:, :, None, None, None]
    return y


def upsample_pad(lo, target_dhw):
    """F.interpolate(x2 trilinear, align_corners=False) (decoder.py:105-106) + F.pad with python floor division
    (segmentation/unet.py:34-40)."""
    x = F.interpolate(lo, scale_factor=2, mode="trilinear", align_corners=False)
    dz = target_dhw[0] - x.shape[2]
    dy = target_dhw[1] - x.shape[3]
    dx = target_dhw[2] - x.shape[4]
    return F.pad(x, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2, dz // 2, dz - dz // 2])


def conv_transpose_pad(x, w, bias, target_dhw):
    """ConvTranspose3d(k3, s2, p1) (decoder.py:99-102) + the same F.pad."""
    y = F.conv_transpose3d(x, w, bias, stride=2, padding=1)
    dz = target_dhw[0] - y.shape[2]
    dy = target_dhw[1] - y.shape[3]
    dx = target_dhw[2] - y.shape[4]
    return F.pad(y, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2, dz // 2, dz - dz // 2])


def dice_loss(logits, target, sigmoid=True, batch=False, squared_pred=False, smooth_nr=1e-5, smooth_dr=1e-5, include_background=True,
              softmax=False, to_onehot_y=False, jaccard=False, weight=None, reduction="mean"):
    """monai.losses.DiceLoss restated (MONAI >= 1.2 is un-vendored and unpinned: requirements.txt:4; PARITY UNPINNED -- the formulas are
    additionally pinned by the hand-worked values of tests/golden/handworked.json). Call site unet3d/scripts/script_utils.py:72; config
    brats2020_config.json:112-116. Order as in MONAI's forward: activation (sigmoid | channel softmax, ignored for one channel), one-hot
    of a label-map target (ignored for one channel), include_background=False drops channel 0 of both tensors, sums, jaccard
    (denominator 2 (D - I)), per-class weight (only for more than one counted class), reduction mean | sum | none
    ("none": [N, C, 1, 1, 1], or [C, 1, 1, 1] with batch)."""
    c = logits.shape[1]
    p = torch.sigmoid(logits) if sigmoid else logits
    if softmax and c > 1:
        p = torch.softmax(p, 1)
    y = target
    if to_onehot_y and c > 1:
        y = torch.zeros_like(p).scatter_(1, target.long(), 1.0)
    y = y.to(p.dtype)
    if not include_background:
        p, y = p[:, 1:], y[:, 1:]
    axes = list(range(2, p.dim()))
    if batch:
        axes = [0] + axes
    inter = torch.sum(y * p, dim=axes)
    if squared_pred:
        g, q = torch.sum(y * y, dim=axes), torch.sum(p * p, dim=axes)
    else:
        g, q = torch.sum(y, dim=axes), torch.sum(p, dim=axes)
    den = g + q
    if jaccard:
        den = 2.0 * (den - inter)
    f = 1.0 - (2.0 * inter + smooth_nr) / (den + smooth_dr)
    if weight is not None and y.shape[1] != 1:
        w = torch.as_tensor(weight, dtype=f.dtype)
        if w.ndim == 0:
            w = w.repeat(y.shape[1])
        f = f * w
    if reduction == "mean":
        return torch.mean(f)
    if reduction == "sum":
        return torch.sum(f)
    return f.view(list(f.shape[0:2]) + [1] * (logits.dim() - 2))


def generalized_dice_loss(logits, target, sigmoid=True, batch=False, smooth_nr=1e-5, smooth_dr=1e-5, include_background=True):
    """monai.losses.GeneralizedDiceLoss(w_type="square", reduction="mean") restated -- the loss doc/Configuration.md:41 configures
    (include_background=False, sigmoid=True). PARITY UNPINNED: MONAI is not importable here; formula as published
    (Sudre et al. 2017) with MONAI's handling of empty classes: an infinite weight is replaced by the largest finite weight of
    the same sample (of the batch with batch=True)."""
    p = torch.sigmoid(logits) if sigmoid else logits
    y = target.to(p.dtype)
    if not include_background:
        p, y = p[:, 1:], y[:, 1:]
    axes = list(range(2, p.dim()))
    if batch:
        axes = [0] + axes
    inter = torch.sum(y * p, dim=axes)
    g, q = torch.sum(y, dim=axes), torch.sum(p, dim=axes)
    w = 1.0 / (g * g)
    infs = torch.isinf(w)
    w = torch.where(infs, torch.zeros_like(w), w)
    if batch:
        w = w + infs * torch.max(w)
        dim = 0
    else:
        w = w + infs * torch.max(w, dim=1, keepdim=True)[0]
        dim = 1
    numer = 2.0 * (inter * w).sum(dim) + smooth_nr
    denom = ((g + q) * w).sum(dim) + smooth_dr
    return torch.mean(1.0 - numer / denom)


def adam_step(p, g, m, v, lr, b1, b2, eps, wd, step):
    """torch.optim.Adam single-tensor update (weight_decay = L2, amsgrad off); script_utils.py:80-81."""
    if wd:
        g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p
