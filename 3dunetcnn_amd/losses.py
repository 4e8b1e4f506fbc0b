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


class _DiceExFunction(torch.autograd.Function):
    """DiceLoss with the options the one-call fused kernel does not carry (softmax, label-map targets, jaccard, weight, reduction):
    forward = sums + finalisation, backward = one pass that applies the upstream gradient of every term."""
    @staticmethod
    def forward(ctx, logits, target, mod, activation):
        be = mod._be or _ops.default_backend(logits.device)
        logits, target = logits.contiguous(), target.contiguous()
        cw = mod.class_weight
        if cw is not None:
            cw = cw.to(device=logits.device, dtype=torch.float32).contiguous()
        loss, state = be.dice_ex_forward(logits, target, activation=activation, batch=mod.batch, squared_pred=mod.squared_pred,
                                         include_background=mod.include_background, jaccard=mod.jaccard, reduction=mod.reduction,
                                         smooth_nr=mod.smooth_nr, smooth_dr=mod.smooth_dr, class_weight=cw)
        ctx.saved = (be, logits, target, state)
        if mod.reduction != "none":
            return loss.reshape(())
        ce = logits.shape[1] - (0 if mod.include_background else 1)
        lead = [ce] if mod.batch else [logits.shape[0], ce]            # MONAI: f.view(list(f.shape[0:2]) + [1] * (input.dim() - 2))
        return loss.reshape(lead + [1] * (logits.dim() - 2))

    @staticmethod
    def backward(ctx, g):
        if ctx.saved is None:
            raise RuntimeError("loss backward called a second time: its saved tensors are released by the first backward")
        be, logits, target, state = ctx.saved
        ctx.saved = None
        return be.dice_ex_backward(logits, target, state, g), None, None, None


class HipDiceLoss(nn.Module):
    """monai.losses.DiceLoss. The shipped configuration (sigmoid / no activation, same-shape targets, reduction="mean") is one fused
    pass (value + gradient); softmax, to_onehot_y, jaccard, weight and reduction="sum" / "none" run the two-call form. other_act (an
    arbitrary Python callable) cannot run inside a kernel and raises."""
    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, other_act=None,
                 squared_pred=False, jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False, weight=None):
        super().__init__()
        if other_act is not None and not callable(other_act):
            raise TypeError(f"other_act must be None or callable but is {type(other_act).__name__}.")       # MONAI's checks, same messages
        if int(sigmoid) + int(softmax) + int(other_act is not None) > 1:
            raise ValueError("Incompatible values: more than 1 of [sigmoid=True, softmax=True, other_act is not None].")
        if other_act is not None:
            raise NotImplementedError(f"{type(self).__name__} does not implement: other_act (a Python callable cannot run inside the fused kernels)")
        reduction = str(getattr(reduction, "value", reduction)).lower()
        if reduction not in ("mean", "sum", "none"):
            raise ValueError(f'Unsupported reduction: {reduction}, available options are ["mean", "sum", "none"].')
        if self.generalized and (to_onehot_y or softmax or jaccard or weight is not None or reduction != "mean"):
            raise NotImplementedError("HipGeneralizedDiceLoss implements sigmoid / no activation, same-shape targets, reduction='mean'")
        self.sigmoid, self.softmax, self.to_onehot_y = bool(sigmoid), bool(softmax), bool(to_onehot_y)
        self.squared_pred, self.jaccard, self.batch = bool(squared_pred), bool(jaccard), bool(batch)
        self.reduction = reduction
        self.smooth_nr = float(smooth_nr)
        self.smooth_dr = float(smooth_dr)
        self.include_background = bool(include_background)
        weight = torch.as_tensor(weight, dtype=torch.float32) if weight is not None else None
        # MONAI raises on a negative weight at every forward. The sign is read once per VALUE of the buffer -- here, while the tensor is
        # still on the host, and again (lazily, in forward) whenever the buffer has been replaced or written since: load_state_dict,
        # `crit.class_weight = ...`, an in-place edit. A device read on every forward would be a host sync inside the step.
        self.register_buffer("class_weight", weight)
        self._weight_checked = None
        self._weight_negative = False
        self._check_weight_sign()
        self._be = None

    generalized = False

    def _check_weight_sign(self):
        """(Re)read the sign of `class_weight` when the buffer is not the one last looked at (identity, storage, version counter)."""
        w = self.class_weight
        if w is None:
            self._weight_checked, self._weight_negative = None, False
            return
        key = (id(w), w.data_ptr(), w._version, w.device)
        if key == self._weight_checked:
            return
        if w.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return                                   # no host read inside a capture: the check happens at the next eager forward
        self._weight_negative = bool(w.numel() and float(w.min()) < 0)
        self._weight_checked = key

    def forward(self, input, target):
        if input.device.type != "cuda" and self._be is None:
            raise RuntimeError(f"{type(self).__name__} runs on an MI355X only (no CPU fallback)")
        c = input.shape[1]
        if not self.include_background and c == 1:
            raise ValueError("single channel prediction, `include_background=False` ignored is not supported: pass include_background=True")
        softmax = self.softmax and c > 1                   # MONAI: "single channel prediction, `softmax=True` ignored."
        onehot = self.to_onehot_y and c > 1                # MONAI: "single channel prediction, `to_onehot_y=True` ignored."
        ce = c - (0 if self.include_background else 1)
        cw = self.class_weight
        if cw is not None and ce != 1:
            if cw.ndim == 0:
                cw = cw.repeat(ce)
            elif cw.shape[0] != ce:
                raise ValueError("the length of the `weight` sequence should be the same as the number of classes. "
                                 "If `include_background=False`, the weight should not include the background category class 0.")
            self._check_weight_sign()
            if self._weight_negative:
                raise ValueError("the value/values of the `weight` should be no less than 0.")
        else:
            cw = None                                        # MONAI applies the weight only for more than one class
        if onehot:
            if target.shape[0] != input.shape[0] or target.shape[1] != 1 or target.shape[2:] != input.shape[2:]:
                raise AssertionError("labels should have a channel with length equal to one.")             # monai.networks.one_hot
            target = target.to(torch.int32)
        else:
            if target.shape != input.shape:
                raise AssertionError(f"ground truth has different shape ({tuple(target.shape)}) from input ({tuple(input.shape)})")
            if target.dtype not in (torch.uint8, torch.float32):
                target = target.to(torch.float32)
        if not (softmax or onehot or self.jaccard or cw is not None or self.reduction != "mean"):
            return _DiceFunction.apply(input.float(), target, self)
        if c > 16:
            raise NotImplementedError("more than 16 classes")
        mod = self if cw is self.class_weight else _WithWeight(self, cw)
        return _DiceExFunction.apply(input.float(), target, mod, "softmax" if softmax else ("sigmoid" if self.sigmoid else None))


class _WithWeight:
    """the module's options with the per-call class weight (a scalar weight is expanded to the number of counted classes per call)"""
    def __init__(self, mod, cw):
        self.__dict__.update({k: getattr(mod, k) for k in ("_be", "batch", "squared_pred", "include_background", "jaccard", "reduction",
                                                           "smooth_nr", "smooth_dr")})
        self.class_weight = cw


def _onehot_u8(be, labels, n_classes):
    """class-index map [N, 1, ...] or [N, ...] -> uint8 one-hot [N, C, ...] on the device (mi355_one_hot, one launch per sample)."""
    if labels.dim() >= 2 and labels.shape[1] == 1 and labels.dim() > 2:
        labels = labels[:, 0]
    groups = [[c] for c in range(n_classes)]
    return torch.stack([be.one_hot(labels[i].float().contiguous(), groups) for i in range(labels.shape[0])])


class _CEFunction(torch.autograd.Function):
    """loss = lambda_dice * Dice + lambda_ce * CE (either weight may be 0), value and d/dlogits from the fused HIP passes."""
    @staticmethod
    def forward(ctx, logits, target, mod):
        be = mod._be or _ops.default_backend(logits.device)
        want = ctx.needs_input_grad[0]
        logits, target = logits.contiguous(), target.contiguous()
        c = logits.shape[1]
        if target.shape != logits.shape:                      # class indices (to_onehot_y / index targets): one-hot once, both terms use it
            target = _onehot_u8(be, target, c)
        loss = dlogits = None
        dice_ex = getattr(mod, "dice_ex", None)              # options the one-call Dice kernel does not carry
        summed = getattr(mod, "reduction", "mean") == "sum"
        if mod.lambda_dice != 0.0:
            if dice_ex is None and not summed:
                loss, dlogits = be.dice(logits, target, sigmoid=mod.sigmoid, batch=mod.batch, squared_pred=mod.squared_pred,
                                        smooth_nr=mod.smooth_nr, smooth_dr=mod.smooth_dr, want_grad=want, grad_scale=mod.lambda_dice,
                                        include_background=mod.include_background)
            else:
                act = "softmax" if (dice_ex or {}).get("softmax") and c > 1 else ("sigmoid" if mod.sigmoid else None)
                loss, state = be.dice_ex_forward(logits, target, activation=act, batch=mod.batch, squared_pred=mod.squared_pred,
                                                 include_background=mod.include_background, jaccard=bool((dice_ex or {}).get("jaccard")),
                                                 reduction="sum" if summed else "mean", smooth_nr=mod.smooth_nr, smooth_dr=mod.smooth_dr)
                if want:
                    dlogits = be.dice_ex_backward(logits, target, state, torch.full((1,), mod.lambda_dice, dtype=torch.float32,
                                                                                    device=logits.device))
            loss.mul_(mod.lambda_dice)
        if mod.lambda_ce != 0.0:
            w = mod.lambda_ce
            if summed:                                        # CrossEntropyLoss / BCEWithLogitsLoss(reduction="sum") = mean x count
                w *= logits.shape[0] * logits[0, 0].numel() * (1 if mod.ce_mode == "softmax" else c)
            loss, dlogits = be.cross_entropy(logits, target, mode=mod.ce_mode, weight=w, loss=loss,
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
        index_shape = (input.shape[0],) + tuple(input.shape[2:])
        as_index = getattr(self, "index_targets", False) and input.shape[1] > 1 and \
            tuple(target.shape) in (index_shape, (input.shape[0], 1) + tuple(input.shape[2:]))
        if target.shape != input.shape and not as_index:
            raise AssertionError(f"ground truth has different shape ({tuple(target.shape)}) from input ({tuple(input.shape)})")
        if input.shape[1] > 16:
            raise NotImplementedError("more than 16 classes")
        if not as_index and target.dtype not in (torch.uint8, torch.float32):
            target = target.to(torch.float32)
        return _CEFunction.apply(input.float(), target, self)


class HipDiceCELoss(_CEBase):
    """monai.losses.DiceCELoss: lambda_dice * DiceLoss(...) + lambda_ce * CrossEntropyLoss(input, target) for more than one channel,
    BCEWithLogitsLoss for a single channel. softmax / jaccard / squared_pred / batch / include_background configure the Dice term,
    to_onehot_y takes a class-index target [N, 1, ...], reduction "mean" | "sum" applies to both terms. Not implemented (raise): `weight`
    (MONAI hands it to the CE term too), `label_smoothing`, `other_act`, reduction "none"."""
    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, other_act=None, squared_pred=False,
                 jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False, weight=None, lambda_dice=1.0,
                 lambda_ce=1.0, label_smoothing=0.0):
        super().__init__()
        reduction = str(getattr(reduction, "value", reduction)).lower()
        if other_act is not None and not callable(other_act):
            raise TypeError(f"other_act must be None or callable but is {type(other_act).__name__}.")
        if int(sigmoid) + int(softmax) + int(other_act is not None) > 1:
            raise ValueError("Incompatible values: more than 1 of [sigmoid=True, softmax=True, other_act is not None].")
        bad = [k for k, v in dict(other_act=other_act is not None, reduction=reduction not in ("mean", "sum"),
                                  weight=weight is not None, label_smoothing=label_smoothing != 0.0).items() if v]
        if bad:
            raise NotImplementedError("HipDiceCELoss does not implement: " + ", ".join(bad))
        # softmax / jaccard go to the Dice term only (MONAI: the CE term always sees the raw logits); to_onehot_y: a class-index target
        # [N, 1, ...] is expanded once on the device and both terms use the expansion (CE of indices == CE of their one-hot)
        self.dice_ex = dict(softmax=bool(softmax), jaccard=bool(jaccard)) if (softmax or jaccard) else None
        self.index_targets = bool(to_onehot_y)
        self.reduction = reduction
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
    """torch.nn.CrossEntropyLoss(reduction="mean") with class-PROBABILITY targets of the input's shape (one-hot uint8 / float) or
    class-INDEX targets [N, ...] (expanded to one-hot on the device).

    `ignore_index` is NOT implemented: torch drops voxels labelled `ignore_index` (default -100) from the sum AND from the mean's
    denominator; here a label outside [0, C) expands to an all-zero one-hot row that contributes 0 to the sum but still counts in the
    denominator, so the value differs from torch's whenever such labels occur. None of the reference's configurations produces them
    (its label maps are re-encoded to {0..C-1} or to nested regions, unet3d/utils/one_hot.py). `validate_targets=True` checks every
    index target for out-of-range labels and raises (one device read = a host sync per call: a debugging aid, off by default)."""
    index_targets = True
    def __init__(self, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction="mean", label_smoothing=0.0,
                 validate_targets=False):
        super().__init__()
        if weight is not None or reduction != "mean" or label_smoothing != 0.0 or size_average is not None or reduce is not None:
            raise NotImplementedError("HipCrossEntropyLoss implements reduction='mean' without weights / smoothing")
        self.ce_mode = "softmax"
        self.validate_targets = bool(validate_targets)
        self._be = None

    def forward(self, input, target):
        if self.validate_targets and target.dim() == input.dim() - 1:
            lo, hi = int(target.min()), int(target.max())
            if lo < 0 or hi >= input.shape[1]:
                raise ValueError(f"HipCrossEntropyLoss: class-index target outside [0, {input.shape[1]}) (min {lo}, max {hi}): "
                                 "ignore_index / out-of-range labels are not implemented")
        return super().forward(input, target)
