"""HipDiceLoss / HipDiceCELoss / HipBCEWithLogitsLoss / HipCrossEntropyLoss: the losses the reference's look-up reaches.

HipDiceLoss: drop-in for monai.losses.DiceLoss as the reference configures it
(examples/brats2020/brats2020_config.json:112-116 -> unet3d/scripts/script_utils.py:61-77, evaluated at
unet3d/train/training_utils.py:111): `criterion(output, target)` -> 0-dim tensor supporting .item() and .backward().

One fused HIP pass computes sigmoid, the three per-(n,c) sums, the loss and d(loss)/d(logits)
(csrc/loss_optim.hip); the target may stay uint8 one-hot (unet3d/transforms/one_hot.py:10).

The cross-entropy leg of the same look-up (script_utils.py:61-77 tries unet3d.losses, torch.nn, monai.losses in that order:
torch.nn.BCEWithLogitsLoss / CrossEntropyLoss, monai.losses.DiceCELoss) is one more fused pass (mi355_ce_fwd_bwd) that adds
its value and gradient to the Dice term's.
"""
import torch
import torch.nn as nn

from . import ops as _ops


class _DiceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, mod):
        be = mod._be or _ops.default_backend(logits.device)
        want = ctx.needs_input_grad[0]
        loss, dlogits = be.dice(logits.contiguous(), target.contiguous(), sigmoid=mod.sigmoid, batch=mod.batch,
                                squared_pred=mod.squared_pred, smooth_nr=mod.smooth_nr, smooth_dr=mod.smooth_dr, want_grad=want,
                                generalized=mod.generalized, include_background=mod.include_background)
        ctx.dlogits = dlogits
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        return _scaled(ctx, g), None, None


def _scaled(ctx, g):
    """d(loss)/d(logits) computed by the forward pass, times the incoming scalar gradient -- in place (no second logits-sized tensor)."""
    d = ctx.dlogits
    if d is None:
        raise RuntimeError("loss backward called a second time: d(loss)/d(logits) is released by the first backward")
    ctx.dlogits = None
    return d.mul_(g)


class HipDiceLoss(nn.Module):
    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, other_act=None,
                 squared_pred=False, jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False, weight=None):
        super().__init__()
        unsupported = []
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
        self.include_background = bool(include_background)
        self._be = None

    generalized = False

    def forward(self, input, target):
        if input.device.type != "cuda" and self._be is None:
            raise RuntimeError(f"{type(self).__name__} runs on an MI355X only (no CPU fallback)")
        if not self.include_background and input.shape[1] == 1:
            raise ValueError("single channel prediction, `include_background=False` ignored is not supported: pass include_background=True")
        if target.shape != input.shape:
            raise AssertionError(f"ground truth has different shape ({tuple(target.shape)}) from input ({tuple(input.shape)})")
        if target.dtype not in (torch.uint8, torch.float32):
            target = target.to(torch.float32)
        return _DiceFunction.apply(input.float(), target, self)


class _CEFunction(torch.autograd.Function):
    """loss = lambda_dice * Dice + lambda_ce * CE (either weight may be 0), value and d/dlogits from the fused HIP passes."""
    @staticmethod
    def forward(ctx, logits, target, mod):
        be = mod._be or _ops.default_backend(logits.device)
        want = ctx.needs_input_grad[0]
        logits, target = logits.contiguous(), target.contiguous()
        loss = dlogits = None
        if mod.lambda_dice != 0.0:
            loss, dlogits = be.dice(logits, target, sigmoid=mod.sigmoid, batch=mod.batch, squared_pred=mod.squared_pred,
                                    smooth_nr=mod.smooth_nr, smooth_dr=mod.smooth_dr, want_grad=want, grad_scale=mod.lambda_dice,
                                    include_background=mod.include_background)
            loss.mul_(mod.lambda_dice)
        if mod.lambda_ce != 0.0:
            loss, dlogits = be.cross_entropy(logits, target, mode=mod.ce_mode, weight=mod.lambda_ce, loss=loss,
                                             dlogits=dlogits if want else None, want_grad=want)
        ctx.dlogits = dlogits
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        return _scaled(ctx, g), None, None


class HipGeneralizedDiceLoss(HipDiceLoss):
    """monai.losses.GeneralizedDiceLoss(w_type="square") -- the loss doc/Configuration.md:41 of the reference configures
    ({"name": "GeneralizedDiceLoss", "include_background": false, "sigmoid": true})."""
    generalized = True

    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, other_act=None, w_type="square",
                 reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False):
        if str(getattr(w_type, "value", w_type)).lower() != "square":
            raise NotImplementedError("HipGeneralizedDiceLoss implements w_type='square' (the MONAI default)")
        super().__init__(include_background=include_background, to_onehot_y=to_onehot_y, sigmoid=sigmoid, softmax=softmax,
                         other_act=other_act, reduction=reduction, smooth_nr=smooth_nr, smooth_dr=smooth_dr, batch=batch)


class _CEBase(nn.Module):
    include_background = True
    sigmoid = True
    squared_pred = batch = False
    smooth_nr = smooth_dr = 1e-5
    lambda_dice, lambda_ce, ce_mode = 0.0, 1.0, "softmax"

    def forward(self, input, target):
        if input.device.type != "cuda" and getattr(self, "_be", None) is None:
            raise RuntimeError(f"{type(self).__name__} runs on an MI355X only (no CPU fallback)")
        if target.shape != input.shape:
            raise AssertionError(f"ground truth has different shape ({tuple(target.shape)}) from input ({tuple(input.shape)})")
        if input.shape[1] > 16:
            raise NotImplementedError("more than 16 classes")
        if target.dtype not in (torch.uint8, torch.float32):
            target = target.to(torch.float32)
        return _CEFunction.apply(input.float(), target, self)


class HipDiceCELoss(_CEBase):
    """monai.losses.DiceCELoss: lambda_dice * DiceLoss(...) + lambda_ce * CrossEntropyLoss(mean)(input, one-hot target as
    probabilities) for more than one channel, BCEWithLogitsLoss(mean) for a single channel."""
    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, other_act=None, squared_pred=False,
                 jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False, weight=None, lambda_dice=1.0,
                 lambda_ce=1.0, label_smoothing=0.0):
        super().__init__()
        bad = [k for k, v in dict(to_onehot_y=to_onehot_y, softmax=softmax,
                                  other_act=other_act is not None, jaccard=jaccard, reduction=reduction != "mean",
                                  weight=weight is not None, label_smoothing=label_smoothing != 0.0).items() if v]
        if bad:
            raise NotImplementedError("HipDiceCELoss does not implement: " + ", ".join(bad))
        self.sigmoid, self.squared_pred, self.batch = bool(sigmoid), bool(squared_pred), bool(batch)
        self.smooth_nr, self.smooth_dr = float(smooth_nr), float(smooth_dr)
        self.lambda_dice, self.lambda_ce = float(lambda_dice), float(lambda_ce)
        if self.lambda_dice < 0.0 or self.lambda_ce < 0.0:
            raise ValueError("lambda_dice and lambda_ce should be no less than 0.0.")      # MONAI's check, same message
        if self.lambda_dice == 0.0 and self.lambda_ce == 0.0:
            raise ValueError("HipDiceCELoss: lambda_dice and lambda_ce are both 0 -- the loss would be identically zero")
        self.include_background = bool(include_background)      # Dice term only (MONAI: the CE term always sees every channel)
        self._be = None

    def forward(self, input, target):
        self.ce_mode = "softmax" if input.shape[1] > 1 else "bce"
        return super().forward(input, target)


class HipBCEWithLogitsLoss(_CEBase):
    """torch.nn.BCEWithLogitsLoss(reduction="mean") on same-shape targets (the multi-label form of the BraTS nested regions)."""
    def __init__(self, weight=None, size_average=None, reduce=None, reduction="mean", pos_weight=None):
        super().__init__()
        if weight is not None or pos_weight is not None or reduction != "mean" or size_average is not None or reduce is not None:
            raise NotImplementedError("HipBCEWithLogitsLoss implements reduction='mean' without weights")
        self.ce_mode = "bce"
        self._be = None


class HipCrossEntropyLoss(_CEBase):
    """torch.nn.CrossEntropyLoss(reduction="mean") with class-PROBABILITY targets of the input's shape (one-hot uint8 / float)."""
    def __init__(self, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction="mean", label_smoothing=0.0):
        super().__init__()
        if weight is not None or reduction != "mean" or label_smoothing != 0.0 or size_average is not None or reduce is not None:
            raise NotImplementedError("HipCrossEntropyLoss implements reduction='mean' without weights / smoothing")
        self.ce_mode = "softmax"
        self._be = None
