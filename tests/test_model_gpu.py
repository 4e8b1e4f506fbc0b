"""Whole-network parity on a real MI355X: HipUNet3D (+ HipDiceLoss, HipAdam) through the C ABI vs the CPU fp32 oracle
(oracle/unet3d_ref.py = restatement of the reference graph, pinned against the reference itself by
tests/test_oracle_pinned.py and tests/golden/). Tolerance 1e-3 relative (BASELINE.json north_star)."""
import importlib
import os

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
optim = importlib.import_module("3dunetcnn_amd.optim")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run_pair(kw, enc, dhw, n, tc=False, seed=1234, train=False):
    """HipUNet3D (+HipDiceLoss) on the GPU vs the CPU oracle graph: logits and loss to TOL against the fp32 oracle; every
    parameter gradient by op_cases.grad_parity (fp32 oracle to TOL, or fp64 oracle to the conditioning-aware allowance).
    Returns `grad` = worst error / allowance (<= 1 passes).
    train=True: the module runs in .train() -- Dropout3d(0.2) after the first block of encoder level 0 (myronenko.py:70-79, 85, 97-100),
    the mode bench.py times -- with a seeded device generator; the keep-mask / (1 - p) the forward drew is read back
    (HipUNet3D.last_dropout_scale) and handed to the oracle graph, so both sides drop the same channels."""
    torch.manual_seed(seed)
    m = unet.HipUNet3D(**kw).cuda().eval()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    x, y = R.synthetic_case(n, kw["n_features"], dhw, kw["n_outputs"])
    ds = None
    if train:
        m.train()
        m.dropout_generator = torch.Generator(device="cuda").manual_seed(77)
        out = m(x.cuda())
        crit = losses.HipDiceLoss(sigmoid=True)
        loss = crit(out, y.cuda())
        loss.backward()
        ds = m.last_dropout_scale.detach().cpu()
        assert ds.shape == (n, kw.get("base_width", 32)) and set(ds.unique().tolist()) == {0.0, 1.25}, ds     # some channels dropped, the rest / 0.8

    def run(dt):
        sd = {k: v.detach().cpu().clone().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
        ref = R.unet3d_forward(sd, x.to(dt), enc, None, tc, dropout_scale=None if ds is None else ds.to(dt))
        l = O.dice_loss(ref, y)
        l.backward()
        return ref.detach(), l.detach(), {k: v.grad for k, v in sd.items()}
    ref, lref, g32 = run(torch.float32)
    _, _, g64 = run(torch.float64)
    floor, perturbed = conditioning.noise_floor(R, lambda: run(torch.float32)[2], return_evals=True)
    if not train:
        out = m(x.cuda())
        crit = losses.HipDiceLoss(sigmoid=True)
        loss = crit(out, y.cuda())
        loss.backward()
    errs = {"logits": C.rel_err(out, ref), "loss": abs(float(loss.detach()) - float(lref)) / abs(float(lref))}
    w = C.grad_parity({k: p.grad for k, p in m.named_parameters()}, g32, g64, floor, TOL, perturbed=perturbed)
    errs["grad"] = w.pop("ratio")
    errs.update(w)
    return errs


# Recorded on MI355X (round 2, exact-fp32 kernels, fused norm statistics; `pytest -s` prints the dictionaries):
#   config                 grad (err/allowance)  n_loose  max_err_vs_fp32   logits
#   32^3 batch 2           0.0014                0        5.2e-5            1.2e-6
#   30x31x29               0.0014                14       8.6e-3            1.2e-6      ragged tiles; ill-conditioned tensors at their floor
#   64^3 (configs[0])      0.099                 3        2.4e-3            1.4e-6
#   transposed convs 32^3  0.0008                0        1.9e-4            1.4e-6
#   five levels 32x48x32   0.0013                0        4.5e-6            8.0e-7
# The asserted bounds leave a factor ~3-5 for summation-order changes; `n_loose` is the count of tensors that miss 1e-3 against the
# fp32 oracle directly and therefore rely on the fp64 / perturbed-oracle legs of op_cases.grad_parity.
# Round 3 (the eligible forward / dgrad convolutions on the Winograd kernel -- a different rounding pattern than the oracle's direct
# oneDNN convolution, each launch still within 1.2e-6 of fp64: tests/test_launch_audit.py -- and train-mode cases added):
#   64^3 (configs[0])      0.146                 20       5.2e-3            1.3e-6
#   five levels 32x48x32   0.070                 0        2.4e-4            6.9e-7
#   train 32^3 batch 2     0.100                 33       4.4e-2            1.1e-6      Dropout3d on: 71 of 90 tensors ill-conditioned
#   train 64^3             0.0095                5        6.9e-3            1.5e-6
# These whole-network gradient numbers are commentary on conditioning; the pass / fail gate of the backward is the
# per-launch fp64 audit, which does not depend on it.
RECORDED = {
    "32": dict(grad=0.02, n_loose=0, max_err_vs_fp32=3e-4, logits=2e-5),
    "odd": dict(grad=0.25, n_loose=18, max_err_vs_fp32=3e-2, logits=2e-5),      # grad 0.11 since the 16^3 level runs on the Winograd kernels too (n_loose 0, max error 7e-4)
    "64": dict(grad=0.5, n_loose=30, max_err_vs_fp32=1.5e-2, logits=2e-5),
    # "tc" since round 5 (conv3d_wino2d_d8, another summation order in the output-transform exchange): the ReLU tie DESIGN section 4 describes
    # -- ONE mask bit of a pre-activation that is zero to fp32 resolution moves decoder.layers.1.blocks.0.conv1.conv.weight by 2.06e-3, the
    # fp32 oracle's own one-ulp runs land on 8e-5 or 2.06e-3 -- now lands on the 2.06e-3 side on the GPU (measured: error 2.0708e-3 against a
    # noise floor of 2.0682e-3, ratio 0.0997, two tensors on the tie leg; before: 0.02 / 0 / 1e-3). The per-launch fp64 audit of the same
    # kernels (tests/test_launch_audit.py) is unchanged at <= 1.2e-6; the bounds sit just above the tie, anything beyond it still fails.
    "tc": dict(grad=0.12, n_loose=2, max_err_vs_fp32=2.2e-3, logits=2e-5),
    "five": dict(grad=0.25, n_loose=2, max_err_vs_fp32=1e-3, logits=2e-5),
    "train32": dict(grad=0.5, n_loose=45, max_err_vs_fp32=1e-1, logits=2e-5),
    "train64": dict(grad=0.1, n_loose=12, max_err_vs_fp32=3e-2, logits=2e-5),
}


@pytest.mark.parametrize("dhw,n,rec", [((32, 32, 32), 2, "32"), ((30, 31, 29), 1, "odd")])
def test_unet3d_default_fwd_bwd(dhw, n, rec):
    e = _run_pair(dict(n_features=4, n_outputs=3), (1, 2, 2, 4), dhw, n)
    print(e)
    assert e["logits"] < TOL and e["loss"] < TOL and e["grad"] <= 1.0, e
    C.assert_recorded(e, RECORDED[rec])


def test_unet3d_default_fwd_bwd_64cube():
    """BASELINE configs[0] size (1x4x64^3)."""
    e = _run_pair(dict(n_features=4, n_outputs=3), (1, 2, 2, 4), (64, 64, 64), 1)
    print(e)
    assert e["logits"] < TOL and e["loss"] < TOL and e["grad"] <= 1.0, e
    C.assert_recorded(e, RECORDED["64"])


@pytest.mark.parametrize("dhw,n,rec", [((32, 32, 32), 2, "train32")])      # (64^3 in train mode: the trajectory test below, step 0)
def test_unet3d_train_mode_dropout_mask_shared_with_oracle(dhw, n, rec):
    """SURVEY 8a row a5 (MyronenkoLayer + Dropout3d) in the mode bench.py times: logits, loss and all gradients of the .train() graph."""
    e = _run_pair(dict(n_features=4, n_outputs=3), (1, 2, 2, 4), dhw, n, train=True)
    print(rec, e)
    assert e["logits"] < TOL and e["loss"] < TOL and e["grad"] <= 1.0, e
    C.assert_recorded(e, RECORDED[rec])


def test_train_mode_loss_trajectory_matches_oracle():
    """bench.py's loop (zero_grad -> forward -> Dice -> backward -> Adam, lr 1e-3, ONE fixed synthetic batch, train mode) for 16 steps at
    64^3 batch 2, beside the oracle graph + torch.optim.Adam fed the SAME Dropout3d masks step by step. The two trajectories must agree
    to 1e-3 while the step-to-step amplification of rounding differences allows it (the first steps), stay close after, and -- the
    question VERDICT r2 raised about bench.py's `final_loss: 1.0` -- saturate together or not at all."""
    torch.manual_seed(1234)
    steps = 12                       # (16 until round 5; both sides collapse at step 10, see below)
    m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().train()
    m.dropout_generator = torch.Generator(device="cuda").manual_seed(5)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    x, y = R.synthetic_case(2, 4, (64, 64, 64), 3)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    opt_ref = torch.optim.Adam(list(sd.values()), lr=1e-3)
    opt = optim.HipAdam(m.parameters(), lr=1e-3)
    crit = losses.HipDiceLoss(sigmoid=True)
    xg, yg = x.cuda(), y.cuda()
    hip, ref = [], []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        l1 = crit(m(xg), yg)
        l1.backward()
        opt.step()
        ds = m.last_dropout_scale.detach().cpu()
        opt_ref.zero_grad()
        l0 = O.dice_loss(R.unet3d_forward(sd, x, dropout_scale=ds), y)
        l0.backward()
        opt_ref.step()
        hip.append(float(l1.detach()))
        ref.append(float(l0.detach()))
    rel = [abs(a - b) / abs(b) for a, b in zip(hip, ref)]
    print("hip   ", [round(v, 5) for v in hip])
    print("oracle", [round(v, 5) for v in ref])
    print("rel   ", [f"{v:.1e}" for v in rel])
    # Recorded on MI355X (round 3): relative difference 6.5e-8, 5.3e-7, 3.0e-6, 3.3e-7, 1.9e-5, 2.6e-4, 3.3e-4, 1.0e-3, 2.2e-3, 1.4e-2 over
    # steps 0-9; at step 10 BOTH sides collapse (0.998 / 0.997: the sigmoid outputs go to 0 everywhere, Dice -> 1) and stay there
    # (1.0 / 0.9972). The collapse is the training problem's (Adam at lr 1e-3 on one noise volume), the oracle does it at the same step:
    # bench.py's `final_loss: 1.0` is not a kernel artefact.
    assert max(rel[:5]) < TOL, (hip, ref)                      # before any amplification: the same step
    assert max(rel) < 5e-2, (hip, ref)                          # the same trajectory
    sat = [v > 0.99 for v in hip], [v > 0.99 for v in ref]
    assert sat[0] == sat[1], (hip, ref)                         # a saturated Dice (p -> 0 everywhere) on one side only would be a bug


def test_unet3d_transposed_conv_variant():
    e = _run_pair(dict(n_features=4, n_outputs=3, use_transposed_convolutions=True), (1, 2, 2, 4), (32, 32, 32), 1, tc=True)
    print(e)
    assert e["logits"] < TOL and e["loss"] < TOL and e["grad"] <= 1.0, e
    C.assert_recorded(e, RECORDED["tc"])


def test_unet3d_five_levels():
    # BASELINE configs[3] topology (encoder_blocks=[1,2,2,2,4], 96.8M params) at a reduced patch so the CPU oracle is quick
    e = _run_pair(dict(n_features=4, n_outputs=3, encoder_blocks=[1, 2, 2, 2, 4]), (1, 2, 2, 2, 4), (32, 48, 32), 1)
    print(e)
    assert e["logits"] < TOL and e["loss"] < TOL and e["grad"] <= 1.0, e
    C.assert_recorded(e, RECORDED["five"])


# The same whole-network case on the DIRECT fp32 kernels at the bounds recorded BEFORE Winograd became the default route (round 2 table
# above): the Winograd bounds in RECORDED are wider because that route rounds differently from the oracle's direct convolution on
# ill-conditioned tensors; a regression of the direct kernels (still the route of stride-2, small-volume and narrow layers, and the
# MI355_WINOGRAD=0 cross-check form) must not hide behind them.
RECORDED_DIRECT = {"five": dict(grad=0.02, n_loose=0, max_err_vs_fp32=5e-5, logits=2e-5),
                   "32": dict(grad=0.02, n_loose=0, max_err_vs_fp32=3e-4, logits=2e-5)}


@pytest.mark.parametrize("rec", ["32"])      # ("five" ran here too until round 6: 7 s of a suite that must stay near 8 minutes; its bounds stay recorded above)
def test_unet3d_direct_kernels_hold_the_round2_bounds(rec, hip_backend):
    old = hip_backend.winograd, hip_backend.wgrad_form
    hip_backend.winograd, hip_backend.wgrad_form = False, "direct"
    try:
        if rec == "five":
            e = _run_pair(dict(n_features=4, n_outputs=3, encoder_blocks=[1, 2, 2, 2, 4]), (1, 2, 2, 2, 4), (32, 48, 32), 1)
        else:
            e = _run_pair(dict(n_features=4, n_outputs=3), (1, 2, 2, 4), (32, 32, 32), 2)
    finally:
        hip_backend.winograd, hip_backend.wgrad_form = old
    print("direct", rec, e)
    assert e["logits"] < TOL and e["loss"] < TOL and e["grad"] <= 1.0, e
    C.assert_recorded(e, RECORDED_DIRECT[rec])


def test_training_steps_match_torch_adam():
    """3 optimizer steps (eval-mode graph so no dropout RNG) of HipUNet3D+HipDiceLoss+HipAdam vs oracle+torch.optim.Adam."""
    torch.manual_seed(7)
    kw = dict(n_features=4, n_outputs=3, base_width=16, encoder_blocks=[1, 1, 2])
    m = unet.HipUNet3D(**kw).cuda().eval()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x, y = R.synthetic_case(2, 4, (32, 32, 32), 3)
    opt_ref = torch.optim.Adam(list(sd.values()), lr=1e-3)
    opt = optim.HipAdam(m.parameters(), lr=1e-3)
    crit = losses.HipDiceLoss(sigmoid=True)
    xg, yg = x.cuda(), y.cuda()
    lh, lr_ = [], []
    for _ in range(3):
        opt_ref.zero_grad()
        l0 = O.dice_loss(R.unet3d_forward(sd, x, (1, 1, 2)), y)
        l0.backward()
        opt_ref.step()
        opt.zero_grad()
        l1 = crit(m(xg), yg)
        l1.backward()
        opt.step()
        lh.append(float(l1))
        lr_.append(float(l0))
    assert all(abs(a - b) / abs(b) < TOL for a, b in zip(lh, lr_)), (lh, lr_)
    # Adam's update is lr * m / (sqrt(v) + eps) ~ lr * sign(g) in the first steps: elements whose gradient is at the
    # fp32 noise floor can take the opposite sign, so compare the bulk of the parameters, not the max.
    tot = bad = 0
    for k, p in m.named_parameters():
        d = (p.detach().cpu() - sd[k].detach()).abs()
        tot += d.numel()
        bad += int((d > 1e-4).sum())
    assert bad / tot < 0.01, (bad, tot)


def test_golden_reference_vectors():
    """Fixture generated in the build container by importing the REFERENCE UNet3D (oracle/make_golden.py)."""
    path = os.path.join(GOLD, "unet3d_small.pt")
    g = torch.load(path)
    m = unet.HipUNet3D(**g["kwargs"]).cuda().eval()
    m.load_state_dict(g["state_dict"])
    out = m(g["x"].cuda())
    crit = losses.HipDiceLoss(sigmoid=True)
    loss = crit(out, g["y"].cuda())
    loss.backward()
    assert C.rel_err(out, g["logits"]) < TOL
    assert abs(float(loss) - float(g["loss"])) / abs(float(g["loss"])) < TOL
    for k, p in m.named_parameters():
        assert C.rel_err(p.grad, g["grads"][k]) < TOL, k


def test_no_grad_inference_and_state_dict_roundtrip(tmp_path):
    torch.manual_seed(3)
    m = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=16, encoder_blocks=[1, 1, 2]).cuda().eval()
    x = torch.randn(1, 4, 24, 24, 24).cuda()
    with torch.no_grad():
        a = m(x)
    torch.save(m.state_dict(), tmp_path / "model.pth")
    m2 = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=16, encoder_blocks=[1, 1, 2]).cuda().eval()
    m2.load_state_dict(torch.load(tmp_path / "model.pth"))
    with torch.no_grad():
        b = m2(x)
    assert torch.equal(a, b)


def test_cpu_input_fails_loudly():
    m = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1])
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 4, 8, 8, 8))


def test_bucketed_allreduce_over_rccl_single_rank():
    """The RCCL leg of the data-parallel path on the one GPU a test box has: a 1-rank `nccl` (= RCCL) process group, bucketed
    asynchronous all-reduces launched from inside the explicit backward, optimizer waiting on them. (World size 2 is covered on
    CPU over gloo in test_ddp_gloo.py; the 8-GPU run is the driver's.)"""
    import socket
    import torch.distributed as dist
    ddp = importlib.import_module("3dunetcnn_amd.ddp")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(9)
        kw = dict(n_features=4, n_outputs=3, base_width=16, encoder_blocks=[1, 1, 2])
        m = unet.HipUNet3D(**kw).cuda().eval()
        m.flatten_parameters()
        red = ddp.GradientBucketReducer(m, bucket_bytes=64 << 10, reduce_single_rank=True)
        red.broadcast_parameters(0)
        x, y = R.synthetic_case(2, 4, (32, 32, 32), 3)
        crit = losses.HipDiceLoss(sigmoid=True)
        opt = optim.HipAdam(m.parameters(), lr=1e-3)
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x.cuda()), y.cuda())
        loss.backward()
        assert red.n_launched == len(red.buckets) >= 3           # every bucket was reduced, launched during backward
        assert red._works == []                                  # ... and joined by the engine at the end of backward (no reducer.wait())
        lref = O.dice_loss(R.unet3d_forward(sd, x, (1, 1, 2)), y)
        lref.backward()
        for k, p in m.named_parameters():
            assert C.rel_err(p.grad, sd[k].grad) < TOL, k        # average over 1 rank == the gradient
        opt.step()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,tol_out,tol_loss", [("bf16", 3e-2, 1e-2), ("fp16", 4e-3, 2e-3)])
def test_autocast_unet_mixed_precision(mode, tol_out, tol_loss):
    """HipAutocastUNet (reference AutocastUNet, unet.py:53-58): 3x3x3 convs on the 16-bit matrix path, fp32 everywhere else.
    bf16 = BASELINE configs[2] 'bf16 mixed precision' (8 significand bits: logits 3e-2, loss 1e-2 vs the fp32 oracle); fp16 = the
    reference class's own CUDA-autocast arithmetic (11 bits: 4e-3 / 2e-3). With fp16 the training steps go through torch's GradScaler,
    as the reference's amp path does (training_utils.py:60-69, 93-96). A few optimizer steps must reduce the loss like the fp32 run does."""
    torch.manual_seed(11)
    kw = dict(n_features=4, n_outputs=3, base_width=16, encoder_blocks=[1, 1, 2])
    m = unet.HipAutocastUNet(autocast_dtype=mode, **kw).cuda().eval()
    assert m.conv_precision == mode and unet.HipAutocastUNet(**kw).conv_precision == "bf16"
    assert unet.HipAutocastUNet(autocast_dtype=torch.float16, **kw).conv_precision == "fp16"
    with pytest.raises(ValueError):
        unet.HipAutocastUNet(autocast_dtype="fp8", **kw)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x, y = R.synthetic_case(2, 4, (32, 32, 32), 3)
    ref = R.unet3d_forward(sd, x, (1, 1, 2))
    lref = O.dice_loss(ref, y)
    crit = losses.HipDiceLoss(sigmoid=True)
    out = m(x.cuda())
    loss = crit(out, y.cuda())
    assert 1e-5 < C.rel_err(out, ref.detach()) < tol_out       # really the 16-bit path (not bit-close), within its tolerance
    assert abs(float(loss) - float(lref)) / float(lref) < tol_loss
    be = importlib.import_module("3dunetcnn_amd.ops").default_backend()
    assert be.precision == 0                                    # the per-module override does not leak into the backend
    opt = optim.HipAdam(m.parameters(), lr=1e-3)
    scaler = torch.amp.GradScaler("cuda") if mode == "fp16" else None
    first = None
    for _ in range(8):
        opt.zero_grad()
        l = crit(m(x.cuda()), y.cuda())
        if scaler is None:
            l.backward()
            opt.step()
        else:
            scaler.scale(l).backward()
            scaler.step(opt)
            scaler.update()
        first = float(l) if first is None else first
    assert float(l) < first


def test_auto_implant_unet_contract():
    torch.manual_seed(2)
    m = unet.HipAutoImplantUNet(n_features=4, n_outputs=4, base_width=8, encoder_blocks=[1, 1]).cuda().eval()
    x = torch.randn(1, 4, 16, 16, 16).cuda()
    with torch.no_grad():
        assert torch.allclose(m(x), m.test(x) - x)


def test_side_stream_weight_gradients_are_bit_identical():
    """The default backward enqueues the weight-gradient kernels on a second HIP stream (engine.py: backward_side_stream) so they
    overlap the HBM-bound norm-backward passes. Same kernels, same inputs, same workspace discipline: the flat gradient must be
    BITWISE the single-stream one, run to run."""
    x, y = R.synthetic_case(2, 4, (48, 40, 32), 3)
    x, y = x.cuda(), y.cuda()
    flat = {}
    for side in (False, True, True):
        torch.manual_seed(5)
        m = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=16).cuda().eval()
        assert m.backward_side_stream is True                     # the shipped default
        m.backward_side_stream = side
        crit = losses.HipDiceLoss(sigmoid=True)
        for _ in range(2):                                        # second pass: allocator blocks are being reused across streams
            m.zero_grad(set_to_none=True)
            crit(m(x), y).backward()
        torch.cuda.synchronize()
        g = m.flat_grad().clone()
        if side in flat:
            assert torch.equal(flat[side], g)
        flat[side] = g
    assert torch.equal(flat[False], flat[True])
