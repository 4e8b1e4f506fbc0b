"""GPU twins of tests/test_act_storage_emu.py: 16-bit activation storage at op level on the MI355X (the HIP library through the C ABI),
at sizes that take the kernels' interior fast paths, several workgroups per CU and ragged edges. tests/act_storage_cases.py states the
property (stored == the fp32-tensor result rounded once)."""
import importlib

import pytest
import torch

import act_storage_cases as S

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dunetcnn_amd.ops")
TOL = S.TOL


@pytest.fixture
def bf16_backend(hip_backend):
    saved = hip_backend.precision
    hip_backend.set_precision("bf16")
    yield hip_backend
    hip_backend.precision = saved


def _all_below(errs, tol=TOL, **special):
    bad = {k: v for k, v in errs.items() if not isinstance(v, bool) and v > special.get(k, tol)}
    assert not bad, (bad, errs)


def test_cast(hip_backend):
    S.case_cast(hip_backend)


def test_pointwise_ops(hip_backend):
    _all_below(S.case_pointwise(hip_backend))


@pytest.mark.parametrize("kw", [dict(n=2, c=32, dhw=(32, 32, 32)), dict(n=1, c=4, dhw=(16, 16, 24), groups=4), dict(c=96, groups=96, slope=0.01, dhw=(8, 8, 8)),
                                dict(n=2, c=256, dhw=(8, 8, 8))])
def test_norm_statistics_and_backward(hip_backend, kw):
    _all_below(S.case_norm(hip_backend, **kw), dgamma=1e-5, dbeta=1e-5)


def test_projection(hip_backend):
    _all_below(S.case_proj(hip_backend), dw=1e-5)


@pytest.mark.parametrize("kw", [dict(), dict(cin=32, cout=64), dict(cin=256, cout=128), dict(cin=64, cout=64, residual=True, dhw=(17, 19, 23)),
                                dict(cin=64, cout=32, residual=True, dhw=(33, 30, 31))])      # conv3d_k1_stream_bf16: many workgroups, ragged last chunk
def test_conv_1x1x1(hip_backend, kw):
    _all_below(S.case_conv_k1(hip_backend, **kw))


@pytest.mark.parametrize("kw", [dict(), dict(cin=64, cout=32, moments=False), dict(cin=128, cout=128)])
def test_conv_stride2(hip_backend, kw):
    _all_below(S.case_conv_s2(hip_backend, **kw), moments=2e-5)


@pytest.mark.parametrize("kw", [dict(), dict(window=True), dict(cin=32, cout=64), dict(cin=32, cout=32), dict(cin=256, cout=128)])
def test_conv_zero_insert(hip_backend, kw):
    _all_below(S.case_conv_zero_insert(hip_backend, **kw))


@pytest.mark.parametrize("kw", [
    dict(cin=32, cout=32, dhw=(16, 32, 32), n=2),                                      # interior tiles, plain input staged without a conversion
    dict(cin=32, cout=32, dhw=(17, 30, 35), norm=True, moments=True),                  # ragged tiles, norm prologue, moment records
    dict(cin=64, cout=32, dhw=(16, 16, 32), norm=True, residual=True, drop=True, moments=True, n=2),
    dict(cin=32, cout=64, dhw=(16, 17, 33), gnb=True, mode=1),                         # dgrad with the norm-backward sums, ragged
    dict(cin=128, cout=128, dhw=(8, 8, 16), norm=True, residual=True, moments=True, n=2),
    dict(cin=256, cout=256, dhw=(8, 8, 16), gnb=True, mode=1),
    dict(cin=40, cout=24, dhw=(7, 9, 19), norm=True),                                  # channel counts that are not tile multiples
])
def test_conv_3x3x3_on_16bit_operands(bf16_backend, kw):
    r = S.case_conv_k3_tile(bf16_backend, **kw)
    if kw.get("gnb"):
        assert r["gnb_fused"]
    _all_below(r, moments=2e-5, gnb=1e-5)


@pytest.mark.parametrize("wide,kw", [
    ("big", dict(cin=128, cout=128, dhw=(8, 12, 32), norm=True, residual=True, moments=True, n=2)),   # 4 x 4 x 16 tile, 4 x 2 accumulator tiles per wave
    ("big", dict(cin=64, cout=256, dhw=(5, 7, 19), mode=1)),                                            # ... ragged, plain input, two channel tiles
    ("small", dict(cin=256, cout=256, dhw=(8, 8, 16), norm=True, moments=True, drop=True, n=2)),       # 2 x 4 x 16 tile, 2 x 2 tiles per wave
    ("small", dict(cin=48, cout=128, dhw=(3, 5, 17), mode=1, residual=True)),
    ("big", dict(cin=128, cout=128, dhw=(8, 8, 32), gnb=True, mode=1, n=2)),                            # norm-backward sums: whole tiles -> the wide form
    ("big", dict(cin=64, cout=128, dhw=(8, 9, 32), gnb=True, mode=1)),                                  # ... ragged -> the 64-channel form
    ("small", dict(cin=256, cout=256, dhw=(6, 8, 16), gnb=True, mode=1, n=2)),                          # ... the small tile keeps 64 channels with these sums
])
def test_conv_3x3x3_wide_tile_forms(bf16_backend, monkeypatch, wide, kw):
    """The 128-output-channel workgroups of the tile kernel (round 6, lp_tile_cfg in csrc/conv3d_bf16.hip): MI355_BF16_WIDE=big / small
    forces them on shapes far below the sizes that select them."""
    monkeypatch.setenv("MI355_BF16_WIDE", wide)
    monkeypatch.setenv("MI355_BF16_FORM", "tile")
    r = S.case_conv_k3_tile(bf16_backend, **kw)
    if kw.get("gnb"):
        assert r["gnb_fused"]
    _all_below(r, moments=2e-5, gnb=1e-5)


@pytest.mark.parametrize("form,kw", [
    ("zring", dict(cin=32, cout=32, dhw=(16, 32, 32), norm=True, moments=True, n=2)),
    ("zring", dict(cin=64, cout=32, dhw=(16, 16, 32), norm=True, residual=True, drop=True, moments=True, n=2)),
    ("zring", dict(cin=32, cout=64, dhw=(9, 16, 32), mode=1)),                                  # plain input (a dgrad): staged as loaded
    ("zring", dict(cin=64, cout=64, dhw=(12, 8, 16), norm=True, residual=True)),
    ("zring1", dict(cin=32, cout=32, dhw=(16, 16, 32), gnb=True, mode=1)),                      # the round-3 ring: norm-backward sums
    ("zring1", dict(cin=24, cout=32, dhw=(5, 8, 16), norm=True, moments=True, residual=True)),
])
def test_conv_3x3x3_plane_ring_forms(bf16_backend, monkeypatch, form, kw):
    """The plane-ring kernels (csrc/conv3d_bf16_zring.hip) on 16-bit tensors: MI355_BF16_FORM=zring forces conv3d_k3_lp_zring2 on every
    eligible shape, zring1 the round-3 kernel."""
    monkeypatch.setenv("MI355_BF16_FORM", form)
    r = S.case_conv_k3_tile(bf16_backend, **kw)
    if kw.get("gnb"):
        assert r["gnb_fused"]
    _all_below(r, moments=2e-5, gnb=1e-5)


@pytest.mark.parametrize("dhw", [(6, 9, 10), (32, 32, 32)])
def test_first_layer(bf16_backend, dhw):
    _all_below(S.case_first_layer(bf16_backend, dhw=dhw), moments=2e-5, wgrad=1e-5, c4bwd_vs_wgrad=1e-5)


@pytest.mark.parametrize("kw", [
    dict(kd=1, stride=1, cin=64, cout=32, dhw=(16, 16, 16), n=2),                      # conv3d_wgrad_k1_lp_tr (2 x 1 tiles)
    dict(kd=1, stride=1, cin=32, cout=64, dhw=(33, 31, 29), n=2),                      # ... 1 x 2 tiles, ragged last chunk, many workgroups
    dict(kd=1, stride=1, cin=128, cout=64, dhw=(17, 16, 19)),                          # ... 4 x 2
    dict(kd=1, stride=1, cin=64, cout=128, dhw=(8, 24, 40), n=2),                      # ... 2 x 4
    dict(kd=3, stride=2, cin=32, cout=32, dhw=(17, 16, 19), n=2),
    dict(kd=3, stride=1, cin=32, cout=32, dhw=(16, 32, 32), norm=True, n=2),           # 16-bit-operand weight gradient
    dict(kd=3, stride=1, cin=64, cout=96, dhw=(9, 15, 19), norm=True),                 # ... its 64-channel workgroup form, ragged
    dict(kd=3, stride=1, cin=256, cout=256, dhw=(8, 8, 16)),
    dict(kd=3, stride=1, cin=32, cout=64, dhw=(19, 21, 37), norm=True, n=2),           # conv3d_wgrad_lp_tr: z chunks, ragged tiles (the three above run it too)
])
def test_weight_gradients(bf16_backend, kw):
    _all_below(S.case_wgrad(bf16_backend, **kw), dw=1e-5)


def test_mixed_storage_types_are_refused(hip_backend):
    be = hip_backend
    a = be.empty_act(1, 4, 4, 8, 32)
    b = be.empty_act(1, 4, 4, 8, 32, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="unsupported"):
        be.add(a, b, a)
    w = be.pack_weight(torch.randn(32, 32, 1, 1, 1, device=be.device), 0)
    with pytest.raises(RuntimeError, match="unsupported"):
        be.conv_fwd(a, w, b, 1)


# ---- IEEE fp16 storage (MI355_ACT_F16, the tensors of the reference's own amp mode) with fp16 operands: GPU twins of the emulator cases ----
@pytest.fixture
def fp16_backend(hip_backend):
    saved = hip_backend.precision
    hip_backend.set_precision("fp16")
    with S.storage_type(torch.float16):
        yield hip_backend
    hip_backend.precision = saved


def test_fp16_cast_pointwise_norm_projection(fp16_backend):
    S.case_cast(fp16_backend)
    _all_below(S.case_pointwise(fp16_backend))
    _all_below(S.case_norm(fp16_backend, n=2, c=32, dhw=(32, 32, 32)), dgamma=1e-5, dbeta=1e-5)
    _all_below(S.case_norm(fp16_backend, c=96, groups=96, slope=0.01, dhw=(8, 8, 8)), dgamma=1e-5, dbeta=1e-5)
    _all_below(S.case_proj(fp16_backend), dw=1e-5)


def test_fp16_conv_1x1x1_stride2_zero_insert(fp16_backend):
    _all_below(S.case_conv_k1(fp16_backend, cin=256, cout=128))
    _all_below(S.case_conv_s2(fp16_backend, cin=128, cout=128), moments=2e-5)
    _all_below(S.case_conv_zero_insert(fp16_backend, window=True))
    _all_below(S.case_conv_zero_insert(fp16_backend, cin=256, cout=128))


@pytest.mark.parametrize("kw", [
    dict(cin=32, cout=32, dhw=(16, 16, 32)),
    dict(cin=64, cout=32, dhw=(8, 8, 32), norm=True, residual=True, drop=True, moments=True, n=2),
    dict(cin=32, cout=64, dhw=(8, 9, 33), gnb=True, mode=1),
    dict(cin=256, cout=256, dhw=(8, 8, 16), norm=True, moments=True),
])
def test_fp16_conv_3x3x3_on_16bit_operands(fp16_backend, kw):
    r = S.case_conv_k3_tile(fp16_backend, **kw)
    if kw.get("gnb"):
        assert r["gnb_fused"]
    _all_below(r, moments=2e-5, gnb=1e-5)


@pytest.mark.parametrize("form,kw", [
    ("zring", dict(cin=32, cout=64, dhw=(16, 32, 32), mode=1)),
    ("zring", dict(cin=64, cout=32, dhw=(12, 16, 32), norm=True, residual=True, drop=True, moments=True, n=2)),
    ("zring", dict(cin=64, cout=64, dhw=(9, 16, 32), norm=True, residual=True)),
    ("zring1", dict(cin=32, cout=32, dhw=(16, 16, 32), gnb=True, mode=1)),
    ("zring1", dict(cin=32, cout=32, dhw=(12, 16, 32), norm=True, moments=True, residual=True)),
])
def test_fp16_conv_3x3x3_plane_ring_forms(fp16_backend, monkeypatch, form, kw):
    monkeypatch.setenv("MI355_BF16_FORM", form)
    r = S.case_conv_k3_tile(fp16_backend, **kw)
    if kw.get("gnb"):
        assert r["gnb_fused"]
    _all_below(r, moments=2e-5, gnb=1e-5)


def test_fp16_first_layer_and_weight_gradients(fp16_backend):
    _all_below(S.case_first_layer(fp16_backend, dhw=(32, 32, 32)), moments=2e-5, wgrad=1e-5, c4bwd_vs_wgrad=1e-5)
    _all_below(S.case_wgrad(fp16_backend, kd=1, stride=1, cin=64, cout=32, dhw=(16, 16, 16)), dw=1e-5)
    _all_below(S.case_wgrad(fp16_backend, kd=3, stride=2, cin=32, cout=32, dhw=(17, 16, 19), n=2), dw=1e-5)
    _all_below(S.case_wgrad(fp16_backend, kd=3, stride=1, cin=32, cout=32, dhw=(9, 10, 34), norm=True), dw=1e-5)
    _all_below(S.case_wgrad(fp16_backend, kd=3, stride=1, cin=64, cout=96, dhw=(6, 8, 33), norm=True), dw=1e-5)


def test_fp16_storage_goes_with_fp16_operands_only(hip_backend):
    be = hip_backend
    saved = be.precision
    try:
        w3 = be.pack_weight(torch.randn(32, 32, 3, 3, 3, device=be.device), 0)
        for prec, dt in (("bf16", torch.float16), ("fp16", torch.bfloat16)):
            be.set_precision(prec)
            x = be.empty_act(1, 4, 4, 16, 32, dtype=dt); x.buf.zero_()
            y = be.empty_act(1, 4, 4, 16, 32, dtype=dt)
            with pytest.raises(RuntimeError, match="unsupported"):
                be.conv_fwd(x, w3, y, 3)
    finally:
        be.precision = saved


def test_bf16_storage_train_steps_track_the_fp32_storage_run():
    """HipAutocastUNet(bf16), 64^3 batch 2: five optimizer steps with activation_storage="bf16" next to the same steps with fp32 tensors
    (same seed, same Dropout3d masks): losses within 2e-3 of each other at every step, and the saved-activation memory about half."""
    unet = importlib.import_module("3dunetcnn_amd.unet"); losses = importlib.import_module("3dunetcnn_amd.losses")
    optim = importlib.import_module("3dunetcnn_amd.optim"); syn = importlib.import_module("3dunetcnn_amd.synthetic")
    x, y = syn.synthetic_case(2, 4, (64, 64, 64))
    x, y = x.cuda(), y.cuda()
    hist, peak = {}, {}
    for st in ("fp32", "bf16"):
        torch.manual_seed(3)
        m = unet.HipAutocastUNet(n_features=4, n_outputs=3, activation_storage=st).cuda().train()
        m.dropout_generator = torch.Generator(device="cuda").manual_seed(11)
        crit = losses.HipDiceLoss(sigmoid=True); opt = optim.HipAdam(m.parameters(), lr=1e-3)
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        ls = []
        for _ in range(5):
            opt.zero_grad(set_to_none=True)
            l = crit(m(x), y); l.backward(); opt.step()
            ls.append(float(l))
        torch.cuda.synchronize()
        hist[st], peak[st] = ls, torch.cuda.max_memory_allocated() - base
        del m, opt
    print(hist, {k: round(v / 2 ** 20) for k, v in peak.items()})
    assert all(abs(a - b) < 2e-3 for a, b in zip(hist["fp32"], hist["bf16"])), hist
    assert peak["bf16"] < 0.7 * peak["fp32"], peak
