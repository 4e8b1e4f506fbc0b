"""HipDiceLoss: drop-in for monai.losses.DiceLoss as the reference configures it
(examples/brats2020/brats2020_config.json:112-116 -> unet3d/scripts/script_utils.py:61-77, evaluated at
unet3d/train/training_utils.py:111): `criterion(output, target)` -> 0-dim tensor supporting .item() and .backward().

One fused HIP pass computes sigmoid, the three per-(n,c) sums, the loss and d(loss)/d(logits)
(csrc/loss_optim.hip); the target may stay uint8 one-hot (unet3d/transforms/one_hot.py:10).
"""
import torch
import torch.nn as nn

from . import ops as _ops


class _DiceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, mod):
        be = mod._be or _ops.default_backend()
        want = ctx.needs_input_grad[0]
        loss, dlogits = be.dice(logits.contiguous(), target.contiguous(), sigmoid=mod.sigmoid, batch=mod.batch,
                                squared_pred=mod.squared_pred, smooth_nr=mod.smooth_nr, smooth_dr=mod.smooth_dr, want_grad=want)
        ctx.dlogits = dlogits
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d = ctx.dlogits
        ctx.dlogits = None
        return d * g, None, None


class HipDiceLoss(nn.Module):
    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, other_act=None,
                 squared_pred=False, jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False, weight=None):
        super().__init__()
        unsupported = []
        if not include_background:
            unsupported.append("include_background=False")
        if to_onehot_y:
            unsupported.append("to_onehot_y=True")
        if softmax:
            unsupported.append("softmax=True")
        if other_act is not None:
            unsupported.append("other_act")
        if jaccard:
            unsupported.append("jaccard=True")
        if reduction != "mean":
            unsupported.append(f"reduction={reduction!r}")
        if weight is not None:
            unsupported.append("weight")
        if unsupported:
            raise NotImplementedError("HipDiceLoss does not implement: " + ", ".join(unsupported))
        self.sigmoid = bool(sigmoid)
        self.squared_pred = bool(squared_pred)
        self.batch = bool(batch)
        self.smooth_nr = float(smooth_nr)
        self.smooth_dr = float(smooth_dr)
        self._be = None

    def forward(self, input, target):
        if input.device.type != "cuda" and self._be is None:
            raise RuntimeError("HipDiceLoss runs on an MI355X only (no CPU fallback)")
        if target.shape != input.shape:
            raise AssertionError(f"ground truth has different shape ({tuple(target.shape)}) from input ({tuple(input.shape)})")
        if target.dtype not in (torch.uint8, torch.float32):
            target = target.to(torch.float32)
        return _DiceFunction.apply(input.float(), target, self)
