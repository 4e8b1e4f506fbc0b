"""HipDynUNet (drop-in for MONAI DynUNet as configured by the reference's BraTS2020 config) vs the plain-torch restatement
oracle/dynunet_ref.py: logits, sigmoid-Dice loss and every parameter gradient. Tolerance 1e-3 relative (north star).
CPU legs run the same kernel sources on the emulator (small filters); GPU legs run the real configuration."""
import importlib
import json
import os

import pytest
import torch

import op_cases as C
from oracle import conditioning, dynunet_ref as D, torch_ops as O, unet3d_ref as R

TOL = 1e-3
dyn = importlib.import_module("3dunetcnn_amd.dynunet")
losses = importlib.import_module("3dunetcnn_amd.losses")
optim = importlib.import_module("3dunetcnn_amd.optim")

BRATS = dict(in_channels=4, out_channels=3, spatial_dims=3, deep_supervision=False,
             strides=[[1, 1, 1]] + [[2, 2, 2]] * 5, filters=[64, 96, 128, 192, 256, 384],
             kernel_size=[[3, 3, 3]] * 6, upsample_kernel_size=[[2, 2, 2]] * 5)   # examples/brats2020/brats2020_config.json:2-107


def _kw(filters):
    L = len(filters)
    return dict(spatial_dims=3, in_channels=4, out_channels=3, kernel_size=[3] * L, strides=[1] + [2] * (L - 1),
                upsample_kernel_size=[2] * (L - 1), filters=filters)


def _pair(m, be, dhw, n, dev):
    """Kernels vs the CPU oracle graph: logits/loss to TOL against the fp32 oracle, gradients by op_cases.grad_parity."""
    L = len(m.filters)
    x, y = R.synthetic_case(n, 4, dhw, 3)
    torch.set_num_threads(min(32, os.cpu_count() or 1))

    def run(dt):
        sd = {k: v.detach().cpu().clone().to(dt).requires_grad_(True) for k, v in m.named_parameters()}
        ref = D.dynunet_forward(sd, x.to(dt), L)
        l = O.dice_loss(ref, y)
        l.backward()
        return ref.detach(), l.detach(), {k: v.grad for k, v in sd.items()}
    ref, lref, g32 = run(torch.float32)
    _, _, g64 = run(torch.float64)
    floor, perturbed = conditioning.noise_floor(D, lambda: run(torch.float32)[2], return_evals=True)
    crit = losses.HipDiceLoss(sigmoid=True)
    if be is not None:
        m._be = be
        crit._be = be
    out = m(x.to(dev))
    loss = crit(out, y.to(dev))
    loss.backward()
    errs = {"logits": C.rel_err(out, ref), "loss": abs(float(loss.detach()) - float(lref)) / abs(float(lref))}
    w = C.grad_parity({k: p.grad for k, p in m.named_parameters()}, g32, g64, floor, TOL, perturbed=perturbed)
    errs["grad"] = w.pop("ratio")
    errs.update(w)
    return errs


def test_state_dict_layout_matches_monai_naming():
    m = dyn.HipDynUNet(**BRATS)
    keys = list(m.state_dict().keys())
    assert keys[:6] == ["input_block.conv1.conv.weight", "input_block.conv2.conv.weight", "input_block.norm1.weight",
                        "input_block.norm1.bias", "input_block.norm2.weight", "input_block.norm2.bias"]
    assert "downsamples.3.conv1.conv.weight" in keys and "bottleneck.conv2.conv.weight" in keys
    assert "upsamples.0.transp_conv.conv.weight" in keys and "upsamples.4.conv_block.norm2.bias" in keys
    own = [k for k in keys if not k.startswith("skip_layers.")]
    assert own[-2:] == ["output_block.conv.conv.weight", "output_block.conv.conv.bias"]
    sd = m.state_dict()
    assert sd["upsamples.0.transp_conv.conv.weight"].shape == (384, 256, 2, 2, 2)
    assert sd["upsamples.4.conv_block.conv1.conv.weight"].shape == (64, 128, 3, 3, 3)
    assert sum(p.numel() for p in m.parameters()) == 24928451
    # MONAI's DynUNet registers every block a second time under skip_layers.* (DynUNetSkipLayer(downsample, next_layer, upsample)
    # chain, innermost next_layer = the bottleneck): state_dict() emits those aliases after output_block, pointing at the SAME tensors,
    # so a checkpoint written here strict-loads into MONAI; loading accepts and ignores them.
    alias = [k for k in keys if k.startswith("skip_layers.")]
    assert keys[len(own):] == alias and len(alias) == len(own) - 2            # everything but the output block is aliased once
    nl = "skip_layers." + "next_layer." * 4
    for a, c in [("skip_layers.downsample.conv1.conv.weight", "input_block.conv1.conv.weight"),
                 ("skip_layers.next_layer.downsample.norm2.bias", "downsamples.0.norm2.bias"),
                 (nl + "downsample.conv2.conv.weight", "downsamples.3.conv2.conv.weight"),
                 (nl + "next_layer.conv1.conv.weight", "bottleneck.conv1.conv.weight"),
                 (nl + "upsample.transp_conv.conv.weight", "upsamples.0.transp_conv.conv.weight"),
                 ("skip_layers.upsample.conv_block.norm1.weight", "upsamples.4.conv_block.norm1.weight")]:
        assert sd[a].data_ptr() == sd[c].data_ptr() and sd[a].shape == sd[c].shape, (a, c)
    m.load_state_dict(sd, strict=True)
    m.load_state_dict({k: v for k, v in sd.items() if not k.startswith("skip_layers.")}, strict=True)     # a checkpoint without them


def test_reference_config_constructs():
    path = "/root/reference/examples/brats2020/brats2020_config.json"
    if not os.path.exists(path):
        pytest.skip("reference tree only exists in the build container")
    cfg = json.load(open(path))["model"]
    name = cfg.pop("name")
    assert name == "DynUNet"
    m = dyn.HipDynUNet(**cfg)                       # the reference passes config["model"] minus "name" verbatim (script_utils.py:51-54)
    assert m.filters == [64, 96, 128, 192, 256, 384]
    with pytest.raises(NotImplementedError):
        dyn.HipDynUNet(**dict(cfg, deep_supervision=True))
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.randn(1, 4, 32, 32, 32))


def test_emulated_dynunet_fwd_bwd(emu_backend):
    torch.manual_seed(5)
    m = dyn.HipDynUNet(**_kw([8, 12, 16])).eval()
    e = _pair(m, emu_backend, (8, 12, 8), 2, "cpu")
    assert e["logits"] < TOL and e["loss"] < TOL and e["grad"] <= 1.0, e
    assert e["max_err_vs_fp32"] < TOL, e         # this small case is well conditioned: plain 1e-3 on every gradient


# Recorded on MI355X (round 2; `pytest -s` prints the dictionaries): grad = worst error / allowance (<= 1 passes), max_err_vs_fp32 =
# worst error against the fp32 oracle over all tensors (the ill-conditioned ones sit at their noise floor: InstanceNorm over 8
# voxels at the 2^3 bottleneck of the BraTS configuration), n_loose = tensors that need the fp64 / perturbed-oracle legs.
#   5 levels 32^3: grad 0.084, n_loose 3, max_err 6.8e-3 | 3 levels 16x24x32 batch 2: 0.110, 3, 4.1e-3 |
#   BraTS config 64^3: 0.122, n_loose 64 of 70, max_err 0.166 -- with a 2^3 bottleneck nearly every gradient of this configuration sits
#   at its conditioning floor ("numerically touchy"); what is held there is the error / allowance ratio.
# n_loose of "five": 6 with the direct and the first Winograd kernels, 8 with conv3d_wino2d_w8, 13 with the 16^3 level on it too (other summation orders; 55 of the 70
# tensors are ill-conditioned here: noise floor 1.8e-2). The conditioning-INDEPENDENT criterion is tests/test_launch_audit.py (every
# launch of the step against fp64 from the same inputs, <= 1e-5).
RECORDED = {"five": dict(grad=0.5, n_loose=16, max_err_vs_fp32=3e-2, logits=5e-5), "three": dict(grad=0.5, n_loose=6, max_err_vs_fp32=2e-2, logits=5e-5),
            "brats": dict(grad=0.5, n_loose=68, max_err_vs_fp32=0.5, logits=5e-5)}


@pytest.mark.gpu
@pytest.mark.parametrize("filters,dhw,n,rec", [([64, 96, 128, 192, 256], (32, 32, 32), 1, "five"), ([32, 64, 96], (16, 24, 32), 2, "three")])
def test_dynunet_fwd_bwd_gpu(filters, dhw, n, rec):
    torch.manual_seed(1234)
    m = dyn.HipDynUNet(**_kw(filters)).cuda().eval()
    e = _pair(m, None, dhw, n, "cuda")
    print(e)
    assert e["logits"] < TOL and e["loss"] < TOL and e["grad"] <= 1.0, e
    C.assert_recorded(e, RECORDED[rec])


@pytest.mark.gpu
def test_dynunet_brats_config_64cube():
    """BASELINE configs[0]: the BraTS2020-config model on a 1x4x64^3 volume."""
    torch.manual_seed(1234)
    m = dyn.HipDynUNet(**BRATS).cuda().eval()
    e = _pair(m, None, (64, 64, 64), 1, "cuda")
    print(e)
    assert e["logits"] < TOL and e["loss"] < TOL and e["grad"] <= 1.0, e
    C.assert_recorded(e, RECORDED["brats"])


@pytest.mark.gpu
def test_dynunet_training_steps():
    torch.manual_seed(3)
    m = dyn.HipDynUNet(**_kw([16, 32, 48])).cuda().train()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.named_parameters()}
    x, y = R.synthetic_case(2, 4, (16, 16, 16), 3)
    opt_ref = torch.optim.Adam(list(sd.values()), lr=1e-3)
    opt = optim.HipAdam(m.parameters(), lr=1e-3)
    crit = losses.HipDiceLoss(sigmoid=True)
    xg, yg = x.cuda(), y.cuda()
    for _ in range(3):
        opt_ref.zero_grad()
        l0 = O.dice_loss(D.dynunet_forward(sd, x, 3), y)
        l0.backward()
        opt_ref.step()
        opt.zero_grad()
        l1 = crit(m(xg), yg)
        l1.backward()
        opt.step()
        assert abs(float(l1) - float(l0)) / abs(float(l0)) < TOL
