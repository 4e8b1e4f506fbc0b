"""Inputs whose channel count is not a multiple of 4 (the reference's default is n_features=1): the network input is carried
zero-padded to a float4 multiple, the first block's parameters as zero-padded copies; logits, loss, every parameter gradient
and the input gradient must match the CPU oracle graph (1e-3 relative, BASELINE north star)."""
import importlib

import pytest
import torch

import op_cases as C
from oracle import conditioning
from oracle import dynunet_ref as D
from oracle import torch_ops as O
from oracle import unet3d_ref as R

unet = importlib.import_module("3dunetcnn_amd.unet")
dyn = importlib.import_module("3dunetcnn_amd.dynunet")
losses = importlib.import_module("3dunetcnn_amd.losses")
TOL = 1e-3


def _check(m, be, dev, fwd_ref, cin, dhw, ref_module=R):
    x, y = R.synthetic_case(1, cin, dhw, 2)

    def run(dt):
        sd = {k: v.detach().cpu().clone().to(dt).requires_grad_(True) for k, v in m.named_parameters()}
        xr = x.clone().to(dt).requires_grad_(True)
        ref = fwd_ref(sd, xr)
        lref = O.dice_loss(ref, y)
        lref.backward()
        g = {k: v.grad for k, v in sd.items()}
        g["__input__"] = xr.grad
        return ref.detach(), float(lref.detach()), g
    ref, lref, g32 = run(torch.float32)
    _, _, g64 = run(torch.float64)
    # conditioning-aware gradient criterion (op_cases.grad_parity): a pre-activation that is zero to fp32 resolution takes either ReLU
    # branch depending on the last bit of the norm statistics -- measured on MI355X in this very configuration (3 input channels,
    # 16x20x24): ONE such tie in the last decoder block moves every upstream gradient by ~2e-5 of the largest gradient, whichever
    # (equally accurate) way the statistics were summed. The probe evaluates both branches of such ties.
    floor, perturbed = conditioning.noise_floor(ref_module, lambda: run(torch.float32)[2], return_evals=True)
    crit = losses.HipDiceLoss(sigmoid=True)
    if be is not None:
        m._be = be
        crit._be = be
    xg = x.to(dev).requires_grad_(True)
    out = m(xg)
    loss = crit(out, y.to(dev))
    loss.backward()
    assert out.shape == ref.shape
    assert C.rel_err(out, ref) < TOL
    assert abs(float(loss.detach()) - lref) / abs(lref) < TOL
    grads = {k: p.grad for k, p in m.named_parameters()}
    grads["__input__"] = xg.grad
    for k, gk in grads.items():
        assert gk.shape == g32[k].shape, k
    # relative to each tensor's own scale, floored at 1e-3 of the largest gradient (a single-channel norm weight has a gradient that
    # is zero up to roundoff: 3e-7 against 1e-1 elsewhere): scale the comparison tensors by that floor
    gmax = max(float(v.abs().max()) for k, v in g32.items() if k != "__input__")
    pad = {k: torch.full((1,), 1e-3 * gmax, dtype=v.dtype) for k, v in g32.items()}

    def padded(d, dt=None):
        return {k: torch.cat((v.detach().cpu().reshape(-1).to(dt or v.dtype), pad[k].to(dt or v.dtype))) for k, v in d.items()}
    w = C.grad_parity(padded(grads, torch.float32), padded(g32), padded(g64), floor, TOL, perturbed=[padded(p_) for p_ in perturbed])
    assert w["ratio"] <= 1.0, w


def _unet(cin):
    torch.manual_seed(3)
    kw = dict(n_features=cin, n_outputs=2, base_width=8, encoder_blocks=[1, 1])
    return unet.HipUNet3D(**kw).eval(), (lambda sd, x: R.unet3d_forward(sd, x, (1, 1)))


def _dyn(cin):
    torch.manual_seed(4)
    m = dyn.HipDynUNet(spatial_dims=3, in_channels=cin, out_channels=2, kernel_size=[3, 3], strides=[1, 2], upsample_kernel_size=[2],
                       filters=[8, 12]).eval()
    return m, (lambda sd, x: D.dynunet_forward(sd, x, 2))


@pytest.mark.parametrize("cin", [1, 3, 6])
def test_unet3d_odd_input_channels_on_emulator(emu_backend, cin):
    m, ref = _unet(cin)
    _check(m, emu_backend, "cpu", ref, cin, (8, 8, 8))


@pytest.mark.parametrize("cin", [1, 3])
def test_dynunet_odd_input_channels_on_emulator(emu_backend, cin):
    m, ref = _dyn(cin)
    _check(m, emu_backend, "cpu", ref, cin, (8, 8, 8), D)


@pytest.mark.gpu
@pytest.mark.parametrize("cin", [1, 3])
def test_odd_input_channels_gpu(cin):
    m, ref = _unet(cin)
    _check(m.cuda(), None, "cuda", ref, cin, (16, 20, 24))
    m, ref = _dyn(cin)
    _check(m.cuda(), None, "cuda", ref, cin, (16, 16, 24), D)


def test_empty_batch_matches_reference_behaviour(emu_backend):
    """N = 0: the reference's Conv3d / GroupNorm stack returns an empty [0, n_outputs, D, H, W] tensor and backward gives
    zero gradients (checked live on /root/reference when it is present); the HIP modules launch nothing and do the same."""
    import importlib
    unet = importlib.import_module("3dunetcnn_amd.unet")
    dyn = importlib.import_module("3dunetcnn_amd.dynunet")
    nets = [unet.HipUNet3D(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1]),
            dyn.HipDynUNet(spatial_dims=3, in_channels=4, out_channels=3, kernel_size=[3] * 3, strides=[1, 2, 2],
                           upsample_kernel_size=[2] * 2, filters=[8, 16, 32])]
    for m in nets:
        m._be = emu_backend
        y = m(torch.zeros(0, 4, 8, 8, 8))
        assert tuple(y.shape) == (0, 3, 8, 8, 8) and y.requires_grad
        y.sum().backward()
        assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in m.parameters())
    from oracle import reference_shim as S
    if S.available():
        ref = S.build_reference_unet3d(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1]).eval()
        assert tuple(ref(torch.zeros(0, 4, 8, 8, 8)).shape) == (0, 3, 8, 8, 8)
