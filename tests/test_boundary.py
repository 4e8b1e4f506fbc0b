"""The drop-in boundary's edges (SURVEY.md 8b "object protocol"): what the reference's builder / loader / training loop may do with
the module besides calling it -- DataParallel wrap, tolerant checkpoint loading, frozen parameters, odd constructor kwargs -- and
that each case either works like the reference or fails loudly, never silently.
Kernels run on the CPU emulator (test infrastructure); /root/reference is imported where the reference's own code is the caller."""
import importlib

import pytest
import torch

from oracle import reference_shim, torch_ops as O, unet3d_ref as R

unet = importlib.import_module("3dunetcnn_amd.unet")
losses = importlib.import_module("3dunetcnn_amd.losses")
optim = importlib.import_module("3dunetcnn_amd.optim")
KW = dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1, 1])


def _model(be, **kw):
    m = unet.HipUNet3D(**{**KW, **kw}).eval()
    m._be = be
    return m


def test_data_parallel_replicas_are_refused_loudly(emu_backend):
    """unet3d/models/build.py:18-20 wraps the model in torch.nn.DataParallel for n_gpus > 1. Its replicas (re-created every forward,
    parameters = broadcast views, torch marks them `_is_replica`) cannot carry the flat parameter / gradient buffers of this engine:
    a replica's forward must raise and name the supported route, not compute garbage."""
    m = _model(emu_backend)
    x, _ = R.synthetic_case(1, 4, (8, 8, 8), 3)
    m(x)                                             # the module itself runs
    rep = m._replicate_for_data_parallel()           # what torch.nn.parallel.replicate creates per device
    assert rep._is_replica
    with pytest.raises(RuntimeError, match="DataParallel.*one process per GPU"):
        rep(x)


@pytest.mark.gpu
def test_data_parallel_wrap_on_one_gpu_runs_the_module_itself(hip_backend):
    """DataParallel over a single device calls the wrapped module directly (no replica): the reference's n_gpus=2 path on a
    one-GPU box, and the unwrap in its loader (build.py:37-41), keep working."""
    m = unet.HipUNet3D(**KW).cuda().eval()
    dp = torch.nn.DataParallel(m, device_ids=[0])
    x, _ = R.synthetic_case(1, 4, (16, 16, 16), 3)
    with torch.no_grad():
        assert torch.equal(dp(x.cuda()), m(x.cuda()))
    dp.module.load_state_dict(m.state_dict(), strict=True)
    if torch.cuda.device_count() == 1:
        with pytest.raises(RuntimeError, match="DataParallel"):
            torch.nn.parallel.replicate(m, [0, 0])[1](x.cuda())


@pytest.mark.skipif(not reference_shim.available(), reason="/root/reference only exists in the build container")
def test_reference_loader_tiles_a_narrower_checkpoint(emu_backend, tmp_path):
    """unet3d/models/build.py:47-64 (match_state_dict_shapes / match_tensor_sizes): with strict=False a checkpoint whose tensors are
    narrower is tiled (torch.cat of copies) and cropped to the model's shapes before load_state_dict. The module must take that
    state_dict like an nn.Module made of torch layers does: same keys, OIDHW shapes, and the flat buffer must follow the loaded values."""
    reference_shim.import_reference_unet()
    importlib.import_module("3dunetcnn_amd.register").register()
    build = importlib.import_module("unet3d.models.build")
    torch.manual_seed(3)
    narrow = unet.HipUNet3D(**{**KW, "n_features": 2, "base_width": 4})         # half the input modalities and half the widths
    path = str(tmp_path / "narrow.pth")
    torch.save(narrow.state_dict(), path)
    model = build.build_or_load_model("HipUNet3D", path, n_gpus=0, strict=False, **KW)
    want = build.match_state_dict_shapes(unet.HipUNet3D(**KW).state_dict(), torch.load(path))
    for k, v in model.state_dict().items():
        assert v.shape == want[k].shape and torch.equal(v, want[k]), k
    w = model.state_dict()["encoder.layers.0.blocks.0.conv1.conv.weight"]       # (8, 4, 3, 3, 3) tiled from (4, 2, 3, 3, 3)
    assert torch.equal(w[:4, :2], w[4:, 2:]) and torch.equal(w[:4, :2], narrow.state_dict()["encoder.layers.0.blocks.0.conv1.conv.weight"])
    # the loaded weights are what the kernels use: forward == oracle on the loaded state_dict
    model.eval()
    model._be = emu_backend
    x, _ = R.synthetic_case(1, 4, (8, 8, 8), 3)
    ref = R.unet3d_forward({k: v.clone() for k, v in model.state_dict().items()}, x, (1, 1, 1))
    out = model(x)
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-3


def test_non_trilinear_interpolation_fails_like_the_reference(emu_backend):
    """The reference accepts any interpolation_mode at construction and passes it to F.interpolate with align_corners=False
    (classification/decoder.py:105-106); torch rejects that at the first forward for every mode but "trilinear". Same here."""
    x, _ = R.synthetic_case(1, 4, (8, 8, 8), 3)
    for mode, exc in (("nearest", ValueError), ("area", ValueError), ("bilinear", NotImplementedError)):
        m = _model(emu_backend, interpolation_mode=mode)                        # construction succeeds, as in the reference
        with pytest.raises(exc) as ei:
            m(x)
        with pytest.raises(exc) as ref:
            torch.nn.functional.interpolate(torch.zeros(1, 1, 2, 2, 2), scale_factor=2, mode=mode, align_corners=False)
        assert str(ei.value) == str(ref.value)
    _model(emu_backend, interpolation_mode="nearest", use_transposed_convolutions=True)(x)      # mode unused with transposed convs
    for kw in (dict(kernel_size=5), dict(downsampling_stride=1), dict(layer_widths=[8, 16, 32])):
        with pytest.raises((NotImplementedError, ValueError)):
            unet.HipUNet3D(**{**KW, **kw})


def test_frozen_parameters_get_no_gradient_and_do_not_move(emu_backend):
    """requires_grad=False (fine-tuning a decoder on a frozen encoder): as with autograd, frozen parameters get no .grad, so an
    optimizer that holds them leaves them alone; the others match the oracle."""
    torch.manual_seed(4)
    m = _model(emu_backend)
    for p in m.encoder.parameters():
        p.requires_grad_(False)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, y = R.synthetic_case(1, 4, (8, 8, 8), 3)
    crit = losses.HipDiceLoss(sigmoid=True)
    crit._be = emu_backend
    opt = optim.HipAdam(m.parameters(), lr=1e-2)
    opt._be = emu_backend
    loss = crit(m(x), y)
    loss.backward()
    sd = {k: v.clone().requires_grad_(not k.startswith("encoder.")) for k, v in before.items()}
    O.dice_loss(R.unet3d_forward(sd, x, (1, 1, 1)), y).backward()
    for k, p in m.named_parameters():
        if k.startswith("encoder."):
            assert p.grad is None, k
        else:
            assert float((p.grad - sd[k].grad).abs().max() / sd[k].grad.abs().max().clamp_min(1e-30)) < 1e-3, k
    opt.step()
    after = m.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before if k.startswith("encoder."))
    assert any(not torch.equal(before[k], after[k]) for k in before if k.startswith("decoder."))


def test_module_surgery_between_steps_is_seen_by_the_engine(emu_backend):
    """The engine keeps its parameter list and its gradient views between steps (host time). What the reference's users do to a built
    model -- swap a Parameter (re-initialised norm), replace a submodule's parameter after a first step, accumulate, zero_grad() either
    way -- must still behave like torch: new Parameters are the ones that are trained and receive .grad, values match the oracle."""
    torch.manual_seed(7)
    m = _model(emu_backend)
    x, y = R.synthetic_case(1, 4, (8, 8, 8), 3)
    crit = losses.HipDiceLoss(sigmoid=True)
    crit._be = emu_backend

    def check(tag):
        for p in m.parameters():
            p.grad = None
        crit(m(x), y).backward()
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        O.dice_loss(R.unet3d_forward(sd, x, (1, 1, 1)), y).backward()
        assert [id(p) for p in m._params()] == [id(p) for p in m.parameters()], tag
        for k, p in m.named_parameters():
            assert p.grad is not None, (tag, k)
            assert float((p.grad - sd[k].grad).abs().max() / sd[k].grad.abs().max().clamp_min(1e-30)) < 1e-3, (tag, k)

    check("as built")
    blk = m.encoder.layers[1].blocks[0]
    blk.conv1.norm1.weight = torch.nn.Parameter(torch.full_like(blk.conv1.norm1.weight, 1.7))       # a swapped Parameter, deep in the tree
    check("after swapping a norm weight")
    m.decoder.layers[-1].blocks[0].conv2.conv.weight = torch.nn.Parameter(torch.randn_like(m.decoder.layers[-1].blocks[0].conv2.conv.weight) * 0.05)
    check("after swapping a conv weight")
    # accumulation over two micro-batches, then zero_grad(set_to_none=False) and a plain step: gradients still match
    crit(m(x), y).backward()
    g2 = {k: p.grad.clone() for k, p in m.named_parameters()}
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    O.dice_loss(R.unet3d_forward(sd, x, (1, 1, 1)), y).backward()
    for k in g2:
        assert float((g2[k] - 2 * sd[k].grad).abs().max() / sd[k].grad.abs().max().clamp_min(1e-30)) < 2e-3, k
    m.zero_grad(set_to_none=False)
    assert all(float(p.grad.abs().max()) == 0.0 for p in m.parameters())
    crit(m(x), y).backward()
    for k, p in m.named_parameters():
        assert float((p.grad - sd[k].grad).abs().max() / sd[k].grad.abs().max().clamp_min(1e-30)) < 1e-3, k


def test_surgery_the_registration_hooks_do_not_see_and_pickling(emu_backend):
    """Writes torch's global registration hooks never fire for -- a direct `_parameters[...]` replacement on a CHILD module -- are caught by
    the engine's once-per-forward walk (engine._check_parameter_cache); a pickled model (torch.save(model)) carries no process-local cache."""
    import pickle
    torch.manual_seed(7)
    m = _model(emu_backend)
    x, y = R.synthetic_case(1, 4, (8, 8, 8), 3)
    crit = losses.HipDiceLoss(sigmoid=True)
    crit._be = emu_backend
    crit(m(x), y).backward()
    norm = m.encoder.layers[1].blocks[0].conv1.norm1
    new = torch.nn.Parameter(torch.full_like(norm.weight, 1.3))
    norm._parameters["weight"] = new                             # no hook fires
    for p in m.parameters():
        p.grad = None
    crit(m(x), y).backward()
    assert [id(p) for p in m._params()] == [id(p) for p in m.parameters()]
    assert new.grad is not None and float(new.grad.abs().max()) > 0.0
    state = m.__getstate__()
    assert "_params_cache" not in state and "_gview_cache" not in state
    assert "_params_cache" in m.__dict__                          # (the live module keeps its caches)
    m.invalidate_parameter_cache()
    assert [id(p) for p in m._params()] == [id(p) for p in m.parameters()]


def test_a_forward_that_raises_leaves_the_shared_backend_as_it_found_it(emu_backend, monkeypatch):
    """HipAutocastUNet switches the backend to its precision and its 16-bit activation storage for the duration of a forward. A forward that
    raises half-way (an OOM the caller catches, a refused call) must undo that: the next fp32 network on the same backend would otherwise
    allocate bf16 tensors and be refused by the fp32 kernels."""
    be = emu_backend
    m = unet.HipAutocastUNet(**KW).eval()
    m._be = be
    x, _ = R.synthetic_case(1, 4, (8, 8, 8), 3)
    real = be.upsample2x_fwd
    monkeypatch.setattr(be, "upsample2x_fwd", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("injected failure in the decoder")))
    with pytest.raises(RuntimeError, match="injected failure"):
        m(x)
    assert be.act_dtype == torch.float32 and be.precision == 0
    monkeypatch.setattr(be, "upsample2x_fwd", real)
    out = _model(be)(x)                                    # an fp32 network right after it runs as if nothing had happened
    assert out.dtype == torch.float32 and torch.isfinite(out).all()
    assert torch.isfinite(m(x)).all() and be.act_dtype == torch.float32      # and so does the mixed-precision one


def test_second_backward_raises_a_clear_error(emu_backend):
    m = _model(emu_backend)
    x, y = R.synthetic_case(1, 4, (8, 8, 8), 3)
    crit = losses.HipDiceLoss(sigmoid=True)
    crit._be = emu_backend
    loss = crit(m(x), y)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time|released"):
        loss.backward()


def test_dice_ce_rejects_degenerate_weights():
    with pytest.raises(ValueError):
        losses.HipDiceCELoss(sigmoid=True, lambda_dice=0.0, lambda_ce=0.0)
    with pytest.raises(ValueError):
        losses.HipDiceCELoss(sigmoid=True, lambda_dice=-1.0)


def _amp_step(dev, be):
    """The reference's `training.amp = true` path (train/train.py:33-37: GradScaler; training_utils.py:60-69, 93-96: autocast around
    model + criterion, scaler.scale(loss).backward(), scaler.step(optimizer), scaler.update()). The HIP modules keep fp32 arithmetic
    under torch's autocast (it only re-types torch ops); what must work is the protocol: a scaled incoming gradient, GradScaler's in-place
    unscale of the .grad views of the flat buffer, its inf check and the optimizer step. One scaled step == one plain step."""
    res = {}
    x, y = R.synthetic_case(1, 4, (16, 16, 16), 3)
    x, y = x.to(dev), y.to(dev)
    for amp in (False, True):
        torch.manual_seed(8)
        m = unet.HipUNet3D(**KW).to(dev).eval()
        crit = losses.HipDiceLoss(sigmoid=True)
        opt = optim.HipAdam(m.parameters(), lr=1e-3)
        if be is not None:
            m._be = crit._be = opt._be = be
        opt.zero_grad()
        if amp:
            scaler = torch.amp.GradScaler(dev)
            with torch.autocast(device_type=dev):
                loss = crit(m(x), y)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            assert scaler.get_scale() >= 65536.0                       # no inf / nan was found: the step was taken, the scale kept
        else:
            loss = crit(m(x), y)
            loss.backward()
            opt.step()
        res[amp] = (float(loss.detach()), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    assert abs(res[True][0] - res[False][0]) < 1e-6
    for k, v in res[False][1].items():
        # Adam's first step is lr * g / (|g| + eps): scaling and unscaling the gradient by 2^16 is exact in fp32
        assert float((res[True][1][k] - v).abs().max()) <= 1e-6, k


@pytest.mark.gpu
def test_grad_scaler_and_autocast_protocol_gpu(hip_backend):
    _amp_step("cuda", None)


def test_grad_scaler_and_autocast_protocol_on_emulator(emu_backend):
    _amp_step("cpu", emu_backend)
