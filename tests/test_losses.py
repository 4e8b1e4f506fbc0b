"""Loss modules the reference's look-up reaches (unet3d/scripts/script_utils.py:61-77): HipDiceCELoss (monai.losses.DiceCELoss),
HipBCEWithLogitsLoss / HipCrossEntropyLoss (the torch.nn fallback), against plain torch on the CPU. Value and input gradient
to 1e-3 relative (measured ~1e-6)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

import op_cases as C
from oracle import torch_ops as O

losses = importlib.import_module("3dunetcnn_amd.losses")
TOL = 1e-3


def _data(n, c, dhw, seed=0):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(n, c, *dhw, generator=g) * 2
    t = C.nested_masks(n, dhw, seed)[:, :c] if c <= 3 else (torch.rand(n, c, *dhw, generator=g) > 0.6).to(torch.uint8)
    return z, t


def _empty(t):
    """Second class empty in sample 0 (infinite generalized-Dice weight -> largest finite weight of the sample)."""
    t = t.clone()
    t[0, 1] = 0
    return t


def _check(crit, ref_fn, be, dev, n, c, dhw, name=""):
    z, t = _data(n, c, dhw)
    if "empty" in name:
        t = _empty(t)
    if "labels" in name:                           # class-index target [N, 1, ...] (to_onehot_y / torch's index form)
        t = torch.randint(0, c, (n, 1) + tuple(dhw), generator=torch.Generator().manual_seed(5))
    zr = z.clone().requires_grad_(True)
    ref = ref_fn(zr, t)
    ref.backward()
    if be is not None:
        crit._be = be
    zg = z.to(dev).requires_grad_(True)
    loss = crit(zg, t.to(dev))
    (loss * 3.0).backward()                     # upstream factor flows through
    assert loss.dim() == 0
    assert abs(float(loss.detach()) - float(ref.detach())) / abs(float(ref.detach())) < TOL
    assert C.rel_err(zg.grad, 3.0 * zr.grad) < TOL


CASES = [
    ("dicece", lambda: losses.HipDiceCELoss(sigmoid=True), lambda z, t: O.dice_loss(z, t, True) + F.cross_entropy(z, t.float()), 3),
    ("dicece_weighted", lambda: losses.HipDiceCELoss(sigmoid=True, lambda_dice=0.3, lambda_ce=2.0, batch=True),
     lambda z, t: 0.3 * O.dice_loss(z, t, True, True) + 2.0 * F.cross_entropy(z, t.float()), 3),
    ("dicece_1ch", lambda: losses.HipDiceCELoss(sigmoid=True), lambda z, t: O.dice_loss(z, t, True) + F.binary_cross_entropy_with_logits(z, t.float()), 1),
    ("dice_nobg", lambda: losses.HipDiceLoss(sigmoid=True, include_background=False), lambda z, t: O.dice_loss(z, t, True, include_background=False), 3),
    ("gdl_doc_config", lambda: losses.HipGeneralizedDiceLoss(include_background=False, sigmoid=True),
     lambda z, t: O.generalized_dice_loss(z, t, True, include_background=False), 4),
    ("gdl_batch", lambda: losses.HipGeneralizedDiceLoss(sigmoid=True, batch=True), lambda z, t: O.generalized_dice_loss(z, t, True, batch=True), 3),
    ("gdl_empty_class", lambda: losses.HipGeneralizedDiceLoss(sigmoid=True), lambda z, t: O.generalized_dice_loss(z, _empty(t), True), 3),
    ("dicece_softmax_labels", lambda: losses.HipDiceCELoss(softmax=True, to_onehot_y=True),
     lambda z, t: O.dice_loss(z, t, False, softmax=True, to_onehot_y=True) + F.cross_entropy(z, t[:, 0]), 4),
    ("dicece_softmax_labels_nobg_weighted", lambda: losses.HipDiceCELoss(softmax=True, to_onehot_y=True, include_background=False,
                                                                          lambda_dice=0.5, lambda_ce=1.5),
     lambda z, t: 0.5 * O.dice_loss(z, t, False, softmax=True, to_onehot_y=True, include_background=False) + 1.5 * F.cross_entropy(z, t[:, 0]), 3),
    ("dicece_jaccard_sum", lambda: losses.HipDiceCELoss(sigmoid=True, jaccard=True, reduction="sum"),
     lambda z, t: O.dice_loss(z, t, True, jaccard=True, reduction="sum") + F.cross_entropy(z, t.float(), reduction="sum"), 3),
    ("dicece_1ch_sum", lambda: losses.HipDiceCELoss(sigmoid=True, reduction="sum"),
     lambda z, t: O.dice_loss(z, t, True, reduction="sum") + F.binary_cross_entropy_with_logits(z, t.float(), reduction="sum"), 1),
    ("ce_labels", lambda: losses.HipCrossEntropyLoss(), lambda z, t: F.cross_entropy(z, t[:, 0]), 5),
    ("bce", lambda: losses.HipBCEWithLogitsLoss(), lambda z, t: F.binary_cross_entropy_with_logits(z, t.float()), 3),
    ("ce", lambda: losses.HipCrossEntropyLoss(), lambda z, t: F.cross_entropy(z, t.float()), 4),
]


@pytest.mark.parametrize("name,mk,ref,c", CASES, ids=[c[0] for c in CASES])
def test_losses_on_emulator(emu_backend, name, mk, ref, c):
    _check(mk(), ref, emu_backend, "cpu", 2, c, (9, 8, 10), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name,mk,ref,c", CASES, ids=[c[0] for c in CASES])
def test_losses_gpu(name, mk, ref, c):
    _check(mk(), ref, None, "cuda", 2, c, (40, 48, 36), name)


# monai DiceLoss options beyond the shipped configuration (the reference passes the config's loss kwargs straight to the class:
# unet3d/scripts/script_utils.py:61-77): (constructor kwargs, channels, label-map target?)
EX_CASES = [
    ("softmax", dict(softmax=True), 4, False),
    ("softmax_onehot_nobg", dict(softmax=True, to_onehot_y=True, include_background=False), 4, True),
    ("onehot_sigmoid", dict(sigmoid=True, to_onehot_y=True), 3, True),
    ("jaccard", dict(sigmoid=True, jaccard=True), 3, False),
    ("jaccard_squared_batch", dict(sigmoid=True, jaccard=True, squared_pred=True, batch=True), 3, False),
    ("weight", dict(sigmoid=True, weight=[0.2, 1.0, 3.0]), 3, False),
    ("weight_scalar_nobg", dict(sigmoid=True, weight=2.5, include_background=False), 3, False),
    ("sum", dict(sigmoid=True, reduction="sum"), 3, False),
    ("none", dict(sigmoid=True, reduction="none"), 3, False),
    ("none_batch_nobg", dict(softmax=True, reduction="none", batch=True, include_background=False), 4, False),
    ("none_softmax_weight_jaccard", dict(softmax=True, reduction="none", weight=[1.0, 0.5, 2.0, 0.0], jaccard=True), 4, False),
    ("no_activation", dict(reduction="sum", jaccard=True), 2, False),
]


def _check_ex(kw, be, dev, n, c, dhw, labels):
    z, t = _data(n, c, dhw, seed=3)
    if labels:
        t = torch.randint(0, c, (n, 1) + tuple(dhw), generator=torch.Generator().manual_seed(5))
    if not (kw.get("sigmoid") or kw.get("softmax")):
        z = torch.sigmoid(z)                                   # "probabilities in": keep the values in the range Dice is meant for
    okw = {("sigmoid" if k == "sigmoid" else k): v for k, v in kw.items()}
    okw.setdefault("sigmoid", False)
    zr = z.clone().requires_grad_(True)
    ref = O.dice_loss(zr, t, **okw)
    up = torch.rand(ref.shape, generator=torch.Generator().manual_seed(7)) + 0.5      # a different upstream gradient for every term
    (ref * up).sum().backward()
    crit = losses.HipDiceLoss(**kw)
    if be is not None:
        crit._be = be
    zg = z.to(dev).requires_grad_(True)
    loss = crit(zg, t.to(dev))
    assert loss.shape == ref.shape
    (loss * up.to(dev)).sum().backward()
    assert C.rel_err(loss.detach().cpu(), ref.detach()) < TOL
    assert C.rel_err(zg.grad.cpu(), zr.grad) < TOL


@pytest.mark.parametrize("name,kw,c,labels", EX_CASES, ids=[c[0] for c in EX_CASES])
def test_dice_options_on_emulator(emu_backend, name, kw, c, labels):
    _check_ex(kw, emu_backend, "cpu", 2, c, (9, 8, 10), labels)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,c,labels", EX_CASES, ids=[c[0] for c in EX_CASES])
def test_dice_options_gpu(name, kw, c, labels):
    _check_ex(kw, None, "cuda", 2, c, (40, 48, 36), labels)


def test_dice_option_validation_matches_monai():
    """monai.losses.DiceLoss.__init__ / forward checks (same exception types; messages restated from MONAI 1.3)."""
    with pytest.raises(ValueError, match="Incompatible values"):
        losses.HipDiceLoss(sigmoid=True, softmax=True)
    with pytest.raises(TypeError, match="other_act must be None or callable"):
        losses.HipDiceLoss(other_act=3)
    with pytest.raises(NotImplementedError, match="other_act"):
        losses.HipDiceLoss(other_act=torch.tanh)
    with pytest.raises(ValueError, match="Unsupported reduction"):
        losses.HipDiceLoss(reduction="median")
    crit = losses.HipDiceLoss(sigmoid=True, weight=[1.0, 2.0])
    crit._be = object()                                        # the checks below run before any kernel
    z, t = torch.zeros(1, 3, 4, 4, 4), torch.zeros(1, 3, 4, 4, 4)
    with pytest.raises(ValueError, match="length of the `weight` sequence"):
        crit(z, t)
    crit = losses.HipDiceLoss(sigmoid=True, weight=[1.0, -2.0, 1.0])
    crit._be = object()
    with pytest.raises(ValueError, match="no less than 0"):
        crit(z, t)
    # a weight that arrives AFTER construction is checked as well: load_state_dict, buffer assignment, an in-place edit
    crit = losses.HipDiceLoss(sigmoid=True, weight=[1.0, 2.0, 1.0])
    crit._be = object()
    crit.load_state_dict({"class_weight": torch.tensor([1.0, -2.0, 1.0])})
    with pytest.raises(ValueError, match="no less than 0"):
        crit(z, t)
    crit = losses.HipDiceLoss(sigmoid=True, weight=[1.0, 2.0, 1.0])
    crit._be = object()
    crit.class_weight = torch.tensor([-1.0, 2.0, 1.0])
    with pytest.raises(ValueError, match="no less than 0"):
        crit(z, t)
    crit = losses.HipDiceLoss(sigmoid=True, weight=[1.0, 2.0, 1.0])
    crit._be = object()
    crit.class_weight[2] = -0.5
    with pytest.raises(ValueError, match="no less than 0"):
        crit(z, t)
    crit = losses.HipDiceLoss(softmax=True, to_onehot_y=True)
    crit._be = object()
    with pytest.raises(AssertionError, match="channel with length equal to one"):
        crit(z, t)


def test_unsupported_options_raise():
    with pytest.raises(NotImplementedError, match="weight"):
        losses.HipDiceCELoss(softmax=True, weight=[1.0, 2.0])
    with pytest.raises(NotImplementedError, match="reduction"):
        losses.HipDiceCELoss(sigmoid=True, reduction="none")
    with pytest.raises(ValueError, match="Incompatible values"):
        losses.HipDiceCELoss(sigmoid=True, softmax=True)
    with pytest.raises(NotImplementedError):
        losses.HipBCEWithLogitsLoss(pos_weight=torch.ones(3))
    with pytest.raises(NotImplementedError):
        losses.HipCrossEntropyLoss(label_smoothing=0.1)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="MI355X"):
            losses.HipBCEWithLogitsLoss()(torch.zeros(1, 3, 4, 4, 4), torch.zeros(1, 3, 4, 4, 4))


def test_cross_entropy_index_targets_out_of_range_are_flagged_on_request():
    """torch's ignore_index is not implemented (losses.py: HipCrossEntropyLoss docstring); `validate_targets=True` turns the silent
    difference into an error before any kernel runs."""
    crit = losses.HipCrossEntropyLoss(validate_targets=True)
    crit._be = object()
    z = torch.zeros(1, 3, 2, 2, 2)
    t = torch.zeros(1, 2, 2, 2, dtype=torch.long)
    t[0, 0, 0, 0] = -100
    with pytest.raises(ValueError, match="ignore_index"):
        crit(z, t)
    t[0, 0, 0, 0] = 3
    with pytest.raises(ValueError, match="outside"):
        crit(z, t)

