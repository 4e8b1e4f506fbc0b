"""DIRECT oracle parity on the headline configuration (BASELINE.json configs[1]: default UNet3D 4ch -> 3cls, 128^3 patch,
batch 2, fp32) -- the exact workload bench.py reports `value` on.

The CPU oracle graph (oracle/unet3d_ref.py, pinned bit-for-bit against the imported reference by tests/test_oracle_pinned.py;
the graph it restates is unet3d/models/pytorch/segmentation/unet.py:27-50) needs ~10 s per 128^3 volume for forward + backward on
the GPU host, so nothing here is indirect: logits, loss, per-class sigmoid-Dice and EVERY parameter gradient of the batch-2 step
are compared element-wise, tolerance 1e-3 relative (max|a-b| / max|b|; north_star).

Gradient criterion. A freshly initialised norm + Dice network has gradients that are small differences of
large cancelling sums: the reference's OWN fp32 CPU gradient is up to 1.1e-3 from its fp64 evaluation at this configuration,
and one-ulp (1e-7) noise on its convolutions moves 67 of the 90 tensors by more than 1e-4 (up to 6.3e-3) -- the per-tensor
`floor` measured by oracle/conditioning.py from four perturbed evaluations of the fp32 oracle. So every tensor must meet

    err(kernel, fp32 oracle) <= max(1e-3, KAPPA * floor)      with KAPPA = 2 here (the small-size tests use 10),

or agree to 1e-3 with one of those perturbed evaluations (a single ReLU tie puts the kernel exactly on the branch some of
them take). The diagnostics that say how much slack was used are ASSERTED against bounds recorded on MI355X (round 2, exact-fp32
kernels, tools/headline_diag.py): 14 tensors above 1e-3 (worst 4.5e-3 = 0.98 x its floor), worst error / allowance 0.63.
A regression that pushes well-conditioned tensors onto the loose leg, or grows the error of the ill-conditioned ones, fails.
Round 3 (forward / dgrad of the eligible layers on the Winograd kernel): logits 1.1e-6, loss 0, Dice 6.8e-9, 15 tensors above 1e-3
(worst 5.0e-3 = 1.1 x its floor), worst error / allowance 0.55 -- the same picture. The conditioning-independent statement about the
backward of this configuration is tests/test_launch_audit.py::test_audit_headline_train_step_gpu (every launch vs fp64, <= 1.2e-6).
"""
import importlib
import os
import time

import pytest
import torch

import op_cases as C
from oracle import conditioning
from oracle import torch_ops as O
from oracle import unet3d_ref as R

pytestmark = pytest.mark.gpu
TOL = 1e-3
unet = importlib.import_module("3dunetcnn_amd.unet")
losses = importlib.import_module("3dunetcnn_amd.losses")

KAPPA = 2.0
# recorded (see the module docstring) -> asserted bound
BOUNDS = {"n_above_tol": (14, 24),            # tensors whose error vs the fp32 oracle exceeds 1e-3 (all of them ill-conditioned)
          "max_err_vs_fp32": (4.5e-3, 1e-2),
          "worst_ratio": (0.63, 1.0),         # max over tensors of err / max(1e-3, KAPPA * floor)
          "logits": (1.3e-6, 1e-4), "loss": (0.0, 1e-5), "dice": (3.6e-9, 1e-5)}


def _soft_dice_per_class(logits, y):
    p = torch.sigmoid(logits.detach().double().cpu())
    t = y.double().cpu()
    dims = (0, 2, 3, 4)
    return 2 * (p * t).sum(dims) / (p.sum(dims) + t.sum(dims))


def _oracle(sd0, x, y, dtype, per_sample):
    """Loss, logits and gradients of the oracle graph. per_sample: evaluate one volume at a time (the network and the Dice
    terms are per-sample, so the batch loss is the mean of the per-sample losses and the gradient the mean of theirs) --
    halves the peak host memory of the fp64 leg."""
    sd = {k: v.detach().cpu().clone().to(dtype).requires_grad_(True) for k, v in sd0.items()}
    n = x.shape[0]
    chunks = [(i, i + 1) for i in range(n)] if per_sample else [(0, n)]
    outs, total = [], 0.0
    for a, b in chunks:
        ref = R.unet3d_forward(sd, x[a:b].to(dtype))
        l = O.dice_loss(ref, y[a:b]) * ((b - a) / n)
        l.backward()
        total += float(l.detach())
        outs.append(ref.detach())
    return torch.cat(outs), total, {k: v.grad for k, v in sd.items()}


def test_headline_config_matches_oracle_directly():
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    torch.manual_seed(1234)                                         # bench.py's weight seed
    m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().eval()    # eval(): Dropout3d is the only stochastic op (SURVEY 8a gotcha 2)
    x, y = R.synthetic_case(2, 4, (128, 128, 128), 3)
    sd0 = m.state_dict()

    out = m(x.cuda())
    loss = losses.HipDiceLoss(sigmoid=True)(out, y.cuda())
    loss.backward()
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    out_c = out.detach().cpu()
    del m, out
    torch.cuda.empty_cache()

    t0 = time.perf_counter()
    ref, lref, g32 = _oracle(sd0, x, y, torch.float32, per_sample=False)
    t32 = time.perf_counter() - t0
    e = {"logits": C.rel_err(out_c, ref), "loss": abs(float(loss.detach()) - lref) / abs(lref),
         "dice": float(((_soft_dice_per_class(out_c, y) - _soft_dice_per_class(ref, y)).abs() / _soft_dice_per_class(ref, y)).max())}
    assert e["logits"] < TOL and e["loss"] < TOL and e["dice"] < TOL, e
    del ref
    # two perturbed evaluations (one per sub-gradient branch of the ties; rounds 2-5 ran four: 100 s of the suite's 20-minute budget for
    # floors that differ by a few per cent -- fewer evaluations can only LOWER a floor, i.e. tighten the criterion)
    floor, perturbed = conditioning.noise_floor(R, lambda: _oracle(sd0, x, y, torch.float32, per_sample=False)[2], seeds=(1, 2), return_evals=True)
    e32 = {k: C.rel_err(grads[k], g32[k]) for k in grads}
    ratio = {}
    for k, v in e32.items():
        r = v / max(TOL, KAPPA * floor[k])
        if r > 1.0:
            r = min(r, min(C.rel_err(grads[k], p[k]) for p in perturbed) / TOL)
        ratio[k] = r
    worst = max(ratio, key=ratio.get)
    e.update(n_above_tol=sum(v > TOL for v in e32.values()), max_err_vs_fp32=max(e32.values()), worst_ratio=ratio[worst], worst_key=worst,
             worst_err=e32[worst], worst_floor=floor[worst], oracle_fp32_s=round(t32, 1))
    print(e)
    for k, (_, bound) in BOUNDS.items():
        assert e[k] <= bound, (k, e)
