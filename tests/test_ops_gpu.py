"""Per-op parity on a real MI355X through the C ABI of libmi355unet3d.so vs the torch-CPU fp32 oracle."""
import pytest

import op_cases as C

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _direct_conv_kernels(hip_backend):
    """These are the op tests of the DIRECT conv kernels: the Winograd routing of Backend.conv_fwd is off here (tests/test_wino_gpu.py
    holds the Winograd kernel to the same cases; tests/test_fullsize_gpu.py compares the two at 128^3)."""
    old, hip_backend.winograd = hip_backend.winograd, False
    yield
    hip_backend.winograd = old
TOL = C.TOL


def ok(r):
    vals = r.values() if isinstance(r, dict) else [r]
    return all(v < TOL for v in vals)


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=8, cout=32, dhw=(6, 7, 9)),
    dict(n=2, cin=4, cout=32, dhw=(32, 32, 32), norm=True),
    dict(n=2, cin=32, cout=32, dhw=(48, 64, 64), norm=True, residual=True, chscale=True),     # 4x8x8 tile config
    dict(n=1, cin=64, cout=32, dhw=(32, 64, 64), norm=True, yld=64, yc0=32),
    dict(n=1, cin=32, cout=32, dhw=(33, 30, 36), stride=2),
    dict(n=2, cin=32, cout=32, dhw=(65, 31, 36), stride=2, xld=64, yld=64, yc0=32),     # conv3d_s2c32_fwd: several z chunks, views of wider buffers
    dict(n=2, cin=64, cout=64, dhw=(16, 16, 16), stride=2),
    dict(n=1, cin=128, cout=256, dhw=(16, 16, 16), norm=True),
    dict(n=2, cin=256, cout=256, dhw=(8, 8, 8), norm=True, residual=True),
    dict(n=1, cin=96, cout=64, dhw=(12, 20, 9), norm=True, groups=96, slope=0.01),
    dict(n=2, cin=256, cout=128, dhw=(16, 16, 16), kd=1),
    dict(n=1, cin=32, cout=64, dhw=(19, 6, 7), kd=1, bias=True),
    dict(n=1, cin=32, cout=96, dhw=(6, 6, 8), xld=64),
])
def test_conv_fwd(hip_backend, kw):
    assert C.case_conv_fwd(hip_backend, **kw) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=64, dhw=(16, 17, 18)),
    dict(n=2, cin=32, cout=32, dhw=(32, 64, 64)),
    dict(n=1, cin=256, cout=256, dhw=(8, 8, 8)),
    dict(n=1, cin=32, cout=32, dhw=(32, 32, 32), stride=2),                    # conv3d_s2c32_dgrad
    dict(n=2, cin=32, cout=32, dhw=(65, 31, 37), stride=2, residual=True),     # ... odd extents, several z chunks, the skip gradient in the epilogue
    dict(n=1, cin=64, cout=64, dhw=(15, 17, 16), stride=2),
])
def test_conv_dgrad(hip_backend, kw):
    assert C.case_conv_dgrad(hip_backend, **kw) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=2, cin=32, cout=32, dhw=(16, 17, 20)),
    dict(n=1, cin=128, cout=64, dhw=(8, 8, 8), pad_to=(16, 16, 16)),
    dict(n=1, cin=40, cout=24, dhw=(5, 6, 7)),
])
def test_transposed_conv_k3s2(hip_backend, kw):
    assert C.case_tconv3(hip_backend, **kw) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=2, cin=4, cout=32, dhw=(33, 40, 47)),      # first-layer dgrad (<= 4 output channels): vector-ALU kernel, ragged tiles
    dict(n=1, cin=4, cout=64, dhw=(16, 24, 32)),      # two 32-channel chunks (DynUNet input-block width)
    dict(n=1, cin=3, cout=40, dhw=(9, 11, 13)),       # 3 output channels, partial second chunk
    dict(n=1, cin=1, cout=12, dhw=(6, 7, 9)),         # odd quad count
])
def test_conv_dgrad_first_layer(hip_backend, kw):
    assert C.case_conv_dgrad(hip_backend, **kw) < TOL


def test_conv_dgrad_first_layer_every_precision_mode(hip_backend):
    # the narrow-output kernel is exact fp32 whatever the backend's arithmetic mode (the fp32 pack is selected in ops.PackedWeight)
    for mode in ("bf16x3", "bf16x6", "bf16", "fp16"):
        hip_backend.set_precision(mode)
        try:
            assert C.case_conv_dgrad(hip_backend, 1, 4, 32, (12, 16, 20)) < TOL
        finally:
            hip_backend.set_precision("fp32")


@pytest.mark.parametrize("kw", [
    dict(n=2, cin=32, cout=32, dhw=(32, 32, 32), norm=True),
    dict(n=1, cin=4, cout=32, dhw=(32, 32, 32), norm=True),
    dict(n=1, cin=64, cout=96, dhw=(15, 15, 19), norm=True, slope=0.01),
    dict(n=2, cin=256, cout=256, dhw=(8, 8, 8), norm=True),
    dict(n=1, cin=32, cout=32, dhw=(33, 32, 36), stride=2),
    dict(n=2, cin=256, cout=64, dhw=(16, 16, 16), kd=1),
    dict(n=2, cin=64, cout=32, dhw=(33, 31, 29), kd=1),                        # conv3d_wgrad_k1_stream, fp32 tensors: 2 x 1 tiles, ragged last chunk
    dict(n=1, cin=32, cout=64, dhw=(24, 24, 40), kd=1),                        # ... 1 x 2
    dict(n=2, cin=64, cout=128, dhw=(8, 24, 40), kd=1),                        # ... 2 x 4 (8-voxel chunks)
    dict(n=1, cin=128, cout=64, dhw=(17, 16, 19), kd=1),                       # ... 4 x 2
])
def test_conv_wgrad(hip_backend, kw):
    assert C.case_conv_wgrad(hip_backend, **kw) < TOL


def test_conv_wgrad_ring_z_chunks(hip_backend):
    # tall thin volume: the plane-ring wgrad cuts each 4x8 column into two z chunks (9 + 8 planes) with ragged y/x tiles
    assert C.case_conv_wgrad(hip_backend, n=1, cin=32, cout=64, dhw=(17, 5, 9), norm=True) < TOL


def test_conv_wgrad_ring_many_columns(hip_backend):
    # more 4x8 columns than workgroups per (co, ci) pair: every workgroup walks several columns into the same accumulators
    assert C.case_conv_wgrad(hip_backend, n=1, cin=32, cout=32, dhw=(8, 96, 190), norm=True) < TOL


@pytest.mark.parametrize("kw", [
    dict(mode="softmax"), dict(mode="bce"), dict(mode="softmax", u8=False, with_dice=True), dict(mode="bce", with_dice=True),
])
def test_cross_entropy(hip_backend, kw):
    assert ok(C.case_ce(hip_backend, 2, 3, (48, 40, 56), **kw))
    assert ok(C.case_ce(hip_backend, 1, 5, (6, 7, 9), **kw))


def test_conv_fwd_mid_tile(hip_backend):
    # 32768 .. 131071 output voxels, > 32 output channels: configuration 8 (4x4x8 tiles, two M tiles per wave), deep K with the
    # two-level accumulation, ragged extents; and the dgrad that runs through the same configuration
    assert C.case_conv_fwd(hip_backend, 2, 128, 128, (32, 32, 32), norm=True, residual=True) < TOL
    assert C.case_conv_fwd(hip_backend, 1, 40, 96, (31, 33, 38)) < TOL
    assert C.case_conv_dgrad(hip_backend, 1, 64, 32, (32, 32, 40)) < TOL


# first-layer (4 input channels) kernels, csrc/conv3d_c4.hip: ragged extents, wide/odd output channel counts, concat slices
@pytest.mark.parametrize("kw", [
    dict(n=1, cin=4, cout=32, dhw=(5, 9, 11)),
    dict(n=2, cin=4, cout=48, dhw=(4, 8, 8), norm=True, slope=0.01, bias=True),
    dict(n=1, cin=4, cout=32, dhw=(7, 6, 10), norm=True, xld=8, yld=64, yc0=32, residual=True, chscale=True),
    dict(n=1, cin=4, cout=8, dhw=(3, 3, 3)),
])
def test_conv_fwd_first_layer(hip_backend, kw):
    assert C.case_conv_fwd(hip_backend, **kw) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=4, cout=32, dhw=(5, 9, 11)),
    dict(n=2, cin=4, cout=48, dhw=(8, 8, 16), norm=True, slope=0.01),
    dict(n=1, cin=4, cout=8, dhw=(3, 3, 3)),
])
def test_conv_wgrad_first_layer(hip_backend, kw):
    assert C.case_conv_wgrad(hip_backend, **kw) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=2, c=32, dhw=(64, 64, 64), groups=8),
    dict(n=2, c=4, dhw=(64, 64, 64), groups=4),
    dict(n=1, c=96, dhw=(8, 8, 8), groups=96, slope=0.01, ld=128),
    dict(n=2, c=256, dhw=(8, 8, 8), groups=8),
])
def test_groupnorm(hip_backend, kw):
    assert ok(C.case_gn(hip_backend, **kw))


# ---- norm statistics fused into the conv epilogues (csrc/gn_fuse.h): every tile configuration the networks use ----
@pytest.mark.parametrize("kw", [
    dict(n=2, cin=32, cout=32, dhw=(64, 64, 64), residual=True, chscale=True),      # 4x8x8 x 32 channels: the 128^3-level kernel
    dict(n=2, cin=32, cout=64, dhw=(64, 64, 64)),                                   # 4x8x8 x 64 channels (4 MFMA tiles per wave)
    dict(n=1, cin=64, cout=128, dhw=(32, 32, 32)),                                  # 4x4x8 x 64, 2x2 wave grid
    dict(n=2, cin=128, cout=256, dhw=(16, 16, 16), residual=True),                  # 2x4x8 tiles, two-level accumulation
    dict(n=2, cin=4, cout=32, dhw=(64, 64, 64)),                                    # first layer (conv3d_c4_fwd)
    dict(n=1, cin=4, cout=64, dhw=(33, 30, 36), groups_out=64),                     # DynUNet input block: ragged, InstanceNorm
    dict(n=2, cin=32, cout=32, dhw=(64, 64, 64), stride=2, norm=False),             # stride-2 down-sampling conv (conv3d_s2c32_fwd)
    dict(n=1, cin=32, cout=32, dhw=(65, 31, 36), stride=2, norm=False),             # ... ragged
    dict(n=1, cin=96, cout=96, dhw=(31, 33, 17), stride=2, groups_out=96),
    dict(n=1, cin=32, cout=32, dhw=(32, 32, 32), yld=64, yc0=32),                   # written into a concat slice
])
def test_conv_epilogue_moments(hip_backend, kw):
    assert C.case_conv_moments(hip_backend, **kw) < 2e-5


def test_concat_statistics_from_two_producers(hip_backend):
    assert C.case_cat_moments(hip_backend, 2, 32, 32, (64, 64, 64)) < 2e-5
    assert C.case_cat_moments(hip_backend, 1, 4, 8, (9, 10, 11)) < 2e-5


@pytest.mark.parametrize("kw", [
    dict(n=2, cin=32, cout=32, dhw=(64, 64, 64)),
    dict(n=2, cin=64, cout=32, dhw=(64, 64, 64)),                                   # 4 MFMA tiles per wave
    dict(n=1, cin=128, cout=128, dhw=(32, 32, 32)),
    dict(n=2, cin=256, cout=256, dhw=(16, 16, 16)),
    dict(n=1, cin=96, cout=64, dhw=(17, 19, 23), groups=96, slope=0.01),            # InstanceNorm + LeakyReLU, ragged
])
def test_norm_backward_sums_from_dgrad_epilogue(hip_backend, kw):
    r = C.case_gn_bwd_fused(hip_backend, **kw)
    assert all(v < 2e-5 for v in r.values()), r


def test_fused_statistics_switch_off(hip_backend):
    be = hip_backend
    be.fused_stats = False
    try:
        assert C.case_conv_moments(be, 1, 32, 32, (16, 16, 16), expect_fused=False) < 2e-5
        r = C.case_gn_bwd_fused(be, 1, 32, 32, (16, 16, 16), expect_fused=False)
        assert all(v < 2e-5 for v in r.values()), r
    finally:
        be.fused_stats = True


@pytest.mark.parametrize("lo,tgt", [((16, 16, 16), (32, 32, 32)), ((8, 7, 9), (15, 13, 17)), ((4, 4, 4), (9, 8, 8))])
def test_upsample(hip_backend, lo, tgt):
    assert ok(C.case_upsample(hip_backend, 2, 32, lo, tgt))


@pytest.mark.parametrize("ratio", [10.0, 30.0])
def test_groupnorm_large_mean(hip_backend, ratio):
    # |mean| >> std (ConvTranspose bias + zero-padded planes produce such groups): shifted-sum statistics and the centred
    # backward must stay at fp32 roundoff; E[x^2]-E[x]^2 loses every digit at ratio 30
    r = C.case_gn(hip_backend, 1, 64, (16, 16, 16), 8, offset=1.7 * ratio)
    assert all(v < 2e-5 for v in r.values()), r


@pytest.mark.parametrize("kw", [dict(n=2, cin=96, cout=64, dhw=(8, 8, 8)), dict(n=1, cin=384, cout=256, dhw=(2, 2, 2)), dict(n=1, cin=32, cout=32, dhw=(5, 6, 7), norm=False, yld=64),
    dict(n=1, cin=128, cout=96, dhw=(16, 16, 16))])
def test_transposed_conv_k2s2(hip_backend, kw):
    assert ok(C.case_tconv2(hip_backend, **kw))


def test_conv_over_concat_with_per_channel_prologue(hip_backend):
    assert ok(C.case_conv_cat_slope(hip_backend, 2, 64, 64, 64, (16, 16, 16)))
    assert ok(C.case_conv_cat_slope(hip_backend, 1, 32, 32, 64, (9, 8, 11), stride=2))


def test_proj(hip_backend):
    assert ok(C.case_proj(hip_backend, 2, 32, 3, (32, 32, 32)))
    assert ok(C.case_proj(hip_backend, 1, 64, 3, (19, 16, 17), bias=True))
    assert ok(C.case_proj(hip_backend, 2, 64, 3, (16, 16, 16), bias=True, norm=True))


def test_dice(hip_backend):
    assert ok(C.case_dice(hip_backend, 2, 3, (64, 64, 64)))
    assert ok(C.case_dice(hip_backend, 2, 3, (32, 32, 32), batch=True, squared=True, u8=False))


def test_adam(hip_backend):
    assert C.case_adam(hip_backend, 1000003) < 1e-5
    assert C.case_adam(hip_backend, 4096, wd=0.01) < 1e-5


def test_layout(hip_backend):
    assert C.case_layout(hip_backend, 2, 4, (33, 32, 31)) == 0.0


# ---- split-bf16 matrix path of the 3x3x3 stride-1 convs (csrc/conv3d_bf16.hip): fp32 in/out, products on bf16 MFMA ----
BF16_TOL = {"bf16x3": 1e-4, "bf16x6": 5e-6, "bf16": 3e-2, "fp16": 4e-3}      # fp16: MI355_PREC_F16, 11 significand bits


@pytest.fixture(params=["bf16x3", "bf16x6", "bf16", "fp16"])
def prec_backend(hip_backend, request):
    hip_backend.set_precision(request.param)
    yield hip_backend, BF16_TOL[request.param]
    hip_backend.set_precision("fp32")


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=32, dhw=(9, 10, 37), norm=True, residual=True, chscale=True),
    dict(n=2, cin=64, cout=96, dhw=(16, 16, 32), norm=True, yld=128, yc0=32),
    dict(n=1, cin=4, cout=32, dhw=(32, 32, 32), norm=True),
    dict(n=2, cin=32, cout=32, dhw=(48, 64, 64), norm=True, residual=True),          # big-tile configuration
    dict(n=1, cin=128, cout=128, dhw=(16, 16, 16)),
    dict(n=1, cin=24, cout=40, dhw=(7, 9, 20), norm=True, slope=0.01, bias=True),
])
def test_conv_fwd_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    assert C.case_conv_fwd(be, **kw) < tol


# the 4-input-channel first layer on the bf16 pipe (conv3d_c4_fwd_bf16)
@pytest.mark.parametrize("kw", [
    dict(n=2, cin=4, cout=32, dhw=(64, 64, 64), norm=True),                                   # 16 z tiles per column, split into chunks
    dict(n=1, cin=4, cout=32, dhw=(33, 30, 37), norm=True, residual=True, chscale=True),      # ragged in every axis
    dict(n=2, cin=4, cout=64, dhw=(16, 24, 32), norm=True, slope=0.01),                       # two co tiles (DynUNet input block)
    dict(n=1, cin=4, cout=48, dhw=(7, 6, 10), bias=True, yld=64, yc0=16),
])
def test_first_layer_fwd_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    assert C.case_conv_fwd(be, **kw) < tol


def test_first_layer_moments_bf16_paths(prec_backend):
    be, tol = prec_backend
    assert C.case_conv_moments(be, 2, 4, 32, (32, 40, 48), ytol=tol, strict_vs_oracle=False) < 2e-5


@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=64, dhw=(16, 17, 18)), dict(n=2, cin=64, cout=32, dhw=(32, 32, 32))])
def test_conv_dgrad_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    assert C.case_conv_dgrad(be, **kw) < tol


@pytest.mark.parametrize("kw", [
    dict(n=2, cin=32, cout=32, dhw=(16, 32, 32), norm=True),
    dict(n=1, cin=4, cout=32, dhw=(32, 32, 32), norm=True),
    dict(n=1, cin=64, cout=96, dhw=(15, 15, 19), norm=True, slope=0.01),
    dict(n=2, cin=128, cout=64, dhw=(8, 8, 16)),
    dict(n=1, cin=32, cout=64, dhw=(64, 64, 64), norm=True),
])
def test_conv_wgrad_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    assert C.case_conv_wgrad(be, **kw) < tol


def test_conv_cat_slope_bf16_paths(prec_backend):
    # per-channel slope table (MONAI DynUNet concat) through the 16-bit kernels; forward (tile form) and weight gradient
    be, tol = prec_backend
    r = C.case_conv_cat_slope(be, 1, 32, 32, 32, (9, 6, 17))
    assert r["fwd"] < tol and r["wgrad"] < tol, r


# ---- plane-ring forms of the 16-bit forward / dgrad kernel (csrc/conv3d_bf16_zring.hip; MI355_BF16_FORM=zring: conv3d_k3_lp_zring2,
#      =zring1: the round-3 kernel): GPU twins of the emulator cases in tests/test_ops_emu.py ----
@pytest.fixture(params=[("bf16", "", "zring"), ("fp16", "", "zring"), ("bf16", "2", "zring"), ("bf16", "5", "zring"), ("bf16", "", "zring1")],
                ids=["bf16", "fp16", "bf16-two-z-ranges", "bf16-five-z-ranges", "v1-bf16"])
def zring_backend(hip_backend, request, monkeypatch):
    monkeypatch.setenv("MI355_BF16_FORM", request.param[2])
    monkeypatch.setenv("MI355_BF16_ZSPLITS", request.param[1])
    hip_backend.set_precision(request.param[0])
    yield hip_backend, BF16_TOL[request.param[0]]
    hip_backend.set_precision("fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=32, dhw=(5, 8, 16), norm=True, residual=True, chscale=True),
    dict(n=2, cin=24, cout=32, dhw=(6, 16, 32), bias=True),
    dict(n=1, cin=32, cout=32, dhw=(1, 8, 16)),
    dict(n=1, cin=20, cout=32, dhw=(9, 8, 32), norm=True, slope=0.01, yld=64, yc0=32),
    dict(n=2, cin=32, cout=32, dhw=(32, 32, 32), norm=True, residual=True),                   # 16 columns x 32 planes
    dict(n=1, cin=32, cout=32, dhw=(4, 8, 16), norm=True),                                    # head + tail steps only
    dict(n=1, cin=32, cout=32, dhw=(22, 8, 16), residual=True),                               # every rotation of the accumulator sets
    dict(n=1, cin=64, cout=32, dhw=(6, 8, 16), norm=True, residual=True, chscale=True),       # channel split over wave pairs
    dict(n=2, cin=64, cout=64, dhw=(8, 16, 16), norm=True, bias=True),                        # two channel tiles
    dict(n=1, cin=48, cout=96, dhw=(7, 8, 32), slope=0.01, norm=True, yld=128, yc0=32),        # padded second channel slice
    dict(n=2, cin=64, cout=64, dhw=(64, 64, 64), norm=True, residual=True),                   # a 64^3-level layer of the headline network
])
def test_conv_fwd_zring_form(zring_backend, kw):
    be, tol = zring_backend
    assert C.case_conv_fwd(be, **kw) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=32, dhw=(6, 8, 16)), dict(n=1, cin=32, cout=64, dhw=(6, 8, 16)), dict(n=2, cin=64, cout=64, dhw=(5, 8, 32))])
def test_conv_dgrad_zring_form(zring_backend, kw):
    be, tol = zring_backend
    assert C.case_conv_dgrad(be, **kw) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=32, dhw=(5, 8, 16), residual=True, chscale=True), dict(n=2, cin=32, cout=32, dhw=(4, 16, 16)),
                                dict(n=1, cin=64, cout=64, dhw=(9, 8, 16), residual=True), dict(n=2, cin=32, cout=32, dhw=(64, 64, 64))])
def test_conv_epilogue_moments_zring_form(zring_backend, kw):
    be, tol = zring_backend
    assert C.case_conv_moments(be, ytol=tol, strict_vs_oracle=False, **kw) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=32, dhw=(5, 8, 16)), dict(n=1, cin=64, cout=32, dhw=(6, 8, 16)), dict(n=1, cin=32, cout=64, dhw=(4, 8, 16))])
def test_norm_backward_sums_zring_form(zring_backend, kw):
    import os
    be, tol = zring_backend
    fused = os.environ["MI355_BF16_FORM"] != "zring" or (16 < kw["cout"] <= 32 and kw["cin"] == 32)      # as in tests/test_ops_emu.py
    r = C.case_gn_bwd_fused(be, compare_unfused=True, expect_fused=fused, **kw)
    assert all(v < 2e-5 for v in r.values()), r


def test_launch_stream_handle_follows_the_current_stream(hip_backend):
    """Backend.stream() hands the library torch's CURRENT stream of the backend's device (raw handle, no Stream object per launch): the
    default stream, a side stream inside torch.cuda.stream(...), and the default one again afterwards -- and a kernel launched inside the
    context is ordered on that stream (its result is visible after synchronising the side stream only)."""
    import torch
    be = hip_backend
    assert be.stream() == torch.cuda.current_stream(be.device).cuda_stream
    s = torch.cuda.Stream(device=be.device)
    x = be.empty_act(1, 4, 4, 4, 8)
    x.buf.fill_(2.0)
    y = be.empty_act(1, 4, 4, 4, 8)
    s.wait_stream(torch.cuda.current_stream(be.device))
    with torch.cuda.stream(s):
        assert be.stream() == s.cuda_stream != 0
        be.add(x, x, y)
    assert be.stream() == torch.cuda.current_stream(be.device).cuda_stream
    s.synchronize()
    assert float(y.buf.min()) == float(y.buf.max()) == 4.0


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(n=1, dhw=(3, 8, 16)),                                   # one column tile, one z chunk
    dict(n=2, dhw=(5, 9, 19), groups=2, slope=0.01),             # ragged tiles, leaky slope, two channels per group
    dict(n=1, dhw=(17, 8, 16), groups=1),                        # two z chunks: the halo planes of a chunk come from its neighbour
    dict(n=1, dhw=(2, 3, 5), xld=8, dyld=64),                    # image smaller than the tile; views of wider buffers
])
def test_first_layer_fused_backward(hip_backend, kw):
    r = C.case_c4_bwd(hip_backend, **kw)
    assert all(v < 1e-4 for v in r.values()), r
