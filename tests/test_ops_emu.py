"""Kernel-logic parity on CPU: the SAME kernel sources compiled against tools/emu (fibers + fmaf-chain MFMA) vs the
torch-CPU oracle. Small shapes only; the real parity gate is test_ops_gpu.py (-m gpu)."""
import os

import pytest
import torch

import op_cases as C

TOL = C.TOL


@pytest.fixture(autouse=True)
def _direct_conv_kernels(emu_backend):
    """Op tests of the DIRECT conv kernels: Winograd routing off (tests/test_wino_emu.py holds the Winograd kernel to the same cases)."""
    old, emu_backend.winograd = emu_backend.winograd, False
    yield
    emu_backend.winograd = old


def ok(r):
    vals = r.values() if isinstance(r, dict) else [r]
    return all(v < TOL for v in vals)


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=8, cout=32, dhw=(6, 7, 9)),
    dict(n=2, cin=4, cout=32, dhw=(5, 8, 8), norm=True),
    dict(n=1, cin=32, cout=32, dhw=(9, 10, 12), stride=2),
    dict(n=2, cin=32, cout=32, dhw=(33, 7, 18), stride=2, xld=64, yld=64, yc0=32),      # conv3d_s2c32_fwd: two z chunks, ragged column tiles, views of wider buffers
    dict(n=2, cin=8, cout=64, dhw=(8, 8, 8), stride=2, norm=True, slope=0.01),
    dict(n=2, cin=64, cout=32, dhw=(5, 6, 7), kd=1),
    dict(n=1, cin=32, cout=64, dhw=(9, 6, 7), kd=1, bias=True),
    dict(n=1, cin=32, cout=32, dhw=(6, 6, 8), norm=True, yld=64, yc0=32, residual=True, chscale=True),
    dict(n=1, cin=32, cout=96, dhw=(6, 6, 8), xld=64),
])
def test_conv_fwd(emu_backend, kw):
    assert C.case_conv_fwd(emu_backend, **kw) < TOL


def test_conv_fwd_mid_tile(emu_backend):
    # 32768 .. 131071 output voxels with > 32 output channels: 4x4x8 tiles, two M tiles per wave (configuration 8)
    assert C.case_conv_fwd(emu_backend, 1, 8, 64, (8, 16, 256), norm=True, residual=True) < TOL
    assert C.case_conv_fwd(emu_backend, 1, 24, 40, (9, 15, 250)) < TOL      # ragged tiles, a partial last channel chunk (24 = 16 + 8)
    assert C.case_conv_dgrad(emu_backend, 1, 64, 8, (8, 16, 256)) < TOL


def test_conv_fwd_big_tile(emu_backend):
    # >= 131072 voxels selects the 4x8x8 / KC=16 configuration used at 128^3
    assert C.case_conv_fwd(emu_backend, 1, 8, 64, (8, 16, 128 * 8), residual=True) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=64, dhw=(6, 7, 8)),
    dict(n=1, cin=32, cout=32, dhw=(8, 8, 8), stride=2),                  # conv3d_s2c32_dgrad: even extents
    dict(n=2, cin=32, cout=32, dhw=(33, 7, 19), stride=2),                # ... odd extents (the last odd plane / row / column has one tap), two z chunks, ragged tiles
    dict(n=1, cin=32, cout=32, dhw=(10, 9, 16), stride=2, residual=True), # ... with the skip gradient added in the epilogue (how the network calls it)
    dict(n=1, cin=64, cout=32, dhw=(6, 8, 10), stride=2, residual=True),  # (the template's residual epilogue)
    dict(n=1, cin=8, cout=32, dhw=(7, 9, 8), stride=2),
    dict(n=1, cin=64, cout=32, dhw=(6, 8, 10), stride=2),     # 64 dx channels: the NT = 2 parity-class configuration
])
def test_conv_dgrad(emu_backend, kw):
    assert C.case_conv_dgrad(emu_backend, **kw) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=2, cin=4, cout=32, dhw=(5, 9, 11)),               # first-layer dgrad: <= 4 output channels -> vector-ALU kernel, ragged tiles
    dict(n=1, cin=4, cout=64, dhw=(4, 8, 8)),                 # two 32-channel chunks (DynUNet input block width)
    dict(n=1, cin=3, cout=40, dhw=(3, 5, 6)),                 # 3 dx channels; 40 dy channels = one full + one partial chunk
    dict(n=1, cin=4, cout=12, dhw=(6, 3, 9)),                 # odd quad count (cinP has a zero-weight pad quad that is never read)
])
def test_conv_dgrad_first_layer(emu_backend, kw):
    assert C.case_conv_dgrad(emu_backend, **kw) < TOL


def test_conv_fwd_narrow_output_forward_pack(emu_backend):
    # the <= 4-output-channel kernel is a plain correlation over any fp32 pack: a forward conv to 3 / 4 channels takes it too
    # (also into a channel slice of a wider buffer); with a prologue or a bias the same problem stays on the generic kernel
    assert C.case_conv_fwd(emu_backend, 1, 16, 3, (5, 6, 7)) < TOL
    assert C.case_conv_fwd(emu_backend, 2, 8, 4, (4, 9, 8), yld=8, yc0=4) < TOL
    assert C.case_conv_fwd(emu_backend, 1, 16, 3, (5, 6, 7), norm=True) < TOL
    assert C.case_conv_fwd(emu_backend, 1, 16, 3, (5, 6, 7), bias=True) < TOL


def test_stride2_dgrad_config_name(emu_backend):
    """The zero-insert data gradient of the 32 -> 32 channel stride-2 convolution launches its own kernel; 64 dx channels stay on the template."""
    import ctypes
    be = emu_backend
    name = ctypes.create_string_buffer(96)
    for cdx, want in ((32, b"conv3d_s2c32_dgrad"), (64, b"conv3d_mfma<")):
        dya, dxa = be.empty_act(1, 4, 4, 5, 32), be.empty_act(1, 8, 7, 9, cdx)
        d = be._desc(3, 1, 1, C.ops.IN_ZERO_INSERT, 0.0, None, None, None, None, None, (0, 0, 0), (8, 7, 9), [])
        xd, yd = dya.desc(), dxa.desc()
        be.lib.mi355_conv3d_fwd_config(ctypes.byref(xd), ctypes.byref(yd), ctypes.byref(d), name, 96)
        assert name.value.startswith(want), name.value


def test_conv_narrow_config_name(emu_backend):
    import ctypes
    be = emu_backend
    xa, ya = be.empty_act(1, 4, 8, 8, 32), be.empty_act(1, 4, 8, 8, 4)
    d = be._desc(3, 1, 1, C.ops.IN_PLAIN, 0.0, None, None, None, None, None, (0, 0, 0), (4, 8, 8), [])
    xd, yd = xa.desc(), ya.desc()
    name = ctypes.create_string_buffer(96)
    be.lib.mi355_conv3d_fwd_config(ctypes.byref(xd), ctypes.byref(yd), ctypes.byref(d), name, 96)
    assert name.value == b"conv3d_c4_dgrad"
    ya8 = be.empty_act(1, 4, 8, 8, 8).desc()
    be.lib.mi355_conv3d_fwd_config(ctypes.byref(xd), ctypes.byref(ya8), ctypes.byref(d), name, 96)
    assert name.value.startswith(b"conv3d_mfma<")


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=32, dhw=(4, 5, 6)),                        # output 7x9x11: ragged parity-class tiles
    dict(n=2, cin=16, cout=64, dhw=(3, 4, 4), pad_to=(6, 8, 8)),      # NT = 2 configuration, F.pad window (one zero plane at the back)
    dict(n=1, cin=40, cout=24, dhw=(2, 6, 5)),                        # partial last channel chunk
])
def test_transposed_conv_k3s2(emu_backend, kw):
    assert C.case_tconv3(emu_backend, **kw) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=2, cin=32, cout=32, dhw=(6, 7, 8)),
    dict(n=1, cin=4, cout=32, dhw=(8, 8, 8), norm=True),
    dict(n=1, cin=64, cout=96, dhw=(5, 5, 9), norm=True, slope=0.01),
    dict(n=1, cin=32, cout=32, dhw=(9, 8, 12), stride=2),                      # conv3d_s2c32_wgrad
    dict(n=2, cin=32, cout=32, dhw=(33, 7, 18), stride=2),                     # ... two z chunks per column, ragged tiles, odd extents
    dict(n=2, cin=64, cout=32, dhw=(5, 6, 7), kd=1),                           # conv3d_wgrad_k1_stream, fp32 tensors: 2 x 1 tiles, ragged last chunk
    dict(n=1, cin=32, cout=64, dhw=(9, 9, 9), kd=1),                           # ... 1 x 2
    dict(n=2, cin=64, cout=128, dhw=(7, 9, 17), kd=1),                         # ... 2 x 4 (8-voxel chunks), several workgroups
    dict(n=1, cin=128, cout=64, dhw=(3, 7, 11), kd=1),                         # ... 4 x 2
    dict(n=1, cin=64, cout=64, dhw=(4, 5, 7), kd=1),                           # (2 x 2 tiles: conv3d_wgrad_mfma<1, 1>)
])
def test_conv_wgrad(emu_backend, kw):
    assert C.case_conv_wgrad(emu_backend, **kw) < TOL


def test_conv_wgrad_ring_z_chunks(emu_backend):
    # tall thin volume: the plane-ring wgrad cuts each 4x8 column into two z chunks (9 + 8 planes) with ragged y/x tiles
    assert C.case_conv_wgrad(emu_backend, n=1, cin=32, cout=64, dhw=(17, 5, 9), norm=True) < TOL


def test_conv_wgrad_ring_many_columns(emu_backend):
    # more 4x8 columns than workgroups per (co, ci) pair: every workgroup walks several columns into the same accumulators
    assert C.case_conv_wgrad(emu_backend, n=1, cin=64, cout=64, dhw=(2, 40, 104), norm=True) < TOL


@pytest.mark.parametrize("kw", [
    dict(mode="softmax"), dict(mode="bce"), dict(mode="softmax", u8=False, with_dice=True), dict(mode="bce", with_dice=True),
])
def test_cross_entropy(emu_backend, kw):
    assert ok(C.case_ce(emu_backend, 2, 3, (12, 12, 12), **kw))
    assert ok(C.case_ce(emu_backend, 1, 5, (6, 7, 9), **kw))


# first-layer (4 input channels) kernels, csrc/conv3d_c4.hip: ragged extents, wide/odd output channel counts, concat slices
@pytest.mark.parametrize("kw", [
    dict(n=1, cin=4, cout=32, dhw=(5, 9, 11)),
    dict(n=2, cin=4, cout=48, dhw=(4, 8, 8), norm=True, slope=0.01, bias=True),
    dict(n=1, cin=4, cout=32, dhw=(7, 6, 10), norm=True, xld=8, yld=64, yc0=32, residual=True, chscale=True),
    dict(n=1, cin=4, cout=8, dhw=(3, 3, 3)),
])
def test_conv_fwd_first_layer(emu_backend, kw):
    assert C.case_conv_fwd(emu_backend, **kw) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=4, cout=32, dhw=(5, 9, 11)),
    dict(n=2, cin=4, cout=48, dhw=(8, 8, 16), norm=True, slope=0.01),
    dict(n=1, cin=4, cout=8, dhw=(3, 3, 3)),
])
def test_conv_wgrad_first_layer(emu_backend, kw):
    assert C.case_conv_wgrad(emu_backend, **kw) < TOL


@pytest.mark.parametrize("kw", [
    dict(n=2, c=32, dhw=(5, 6, 7), groups=8),
    dict(n=2, c=4, dhw=(5, 6, 7), groups=4),
    dict(n=1, c=96, dhw=(4, 4, 4), groups=96, slope=0.01, ld=128),
])
def test_groupnorm(emu_backend, kw):
    assert ok(C.case_gn(emu_backend, **kw))


# ---- norm statistics fused into conv epilogues (csrc/gn_fuse.h): index logic of every tile configuration on the emulator ----
@pytest.mark.parametrize("kw", [
    dict(n=2, cin=8, cout=32, dhw=(5, 6, 9)),                                   # ragged tiles, 2 samples (cfg 7: 4x4x8 tiles)
    dict(n=1, cin=16, cout=64, dhw=(3, 5, 8), residual=True, chscale=True),     # 2x2 wave grid, residual with |mean| >> std, dropout scale
    dict(n=1, cin=8, cout=40, dhw=(4, 4, 8), groups_out=40),                    # partial channel tile, InstanceNorm-style groups
    dict(n=1, cin=4, cout=32, dhw=(5, 9, 9)),                                   # first-layer kernel (conv3d_c4_fwd)
    dict(n=1, cin=32, cout=32, dhw=(9, 8, 8), stride=2, norm=False),            # stride-2 down-sampling conv (conv3d_s2c32_fwd)
    dict(n=2, cin=32, cout=32, dhw=(33, 7, 18), stride=2, norm=False),          # ... two z chunks per column, ragged tiles
    dict(n=1, cin=8, cout=32, dhw=(4, 4, 8), yld=64, yc0=32),                   # output written into a concat slice
])
def test_conv_epilogue_moments(emu_backend, kw):
    assert C.case_conv_moments(emu_backend, **kw) < 2e-5


def test_conv_epilogue_moments_large_tiles(emu_backend):
    # >= 131072 output voxels: the 4x8x8 two-M-tile configurations (cfg 4 / 5) of the 128^3 layers
    assert C.case_conv_moments(emu_backend, 1, 8, 64, (32, 64, 64), norm=False) < 2e-5


def test_concat_statistics_from_two_producers(emu_backend):
    assert C.case_cat_moments(emu_backend, 2, 8, 24, (3, 5, 8)) < 2e-5
    assert C.case_cat_moments(emu_backend, 1, 4, 8, (4, 4, 4)) < 2e-5          # 12 channels: per-channel groups


@pytest.mark.parametrize("kw", [
    dict(n=2, cin=32, cout=32, dhw=(5, 6, 9)),
    dict(n=1, cin=64, cout=16, dhw=(3, 5, 8), slope=0.01),                      # 2 N tiles per wave, LeakyReLU
    dict(n=1, cin=40, cout=8, dhw=(4, 4, 8), groups=40),                        # partial channel tile, InstanceNorm
])
def test_norm_backward_sums_from_dgrad_epilogue(emu_backend, kw):
    r = C.case_gn_bwd_fused(emu_backend, **kw)
    assert all(v < 2e-5 for v in r.values()), r


def test_fused_statistics_switch_off(emu_backend):
    """Backend.fused_stats = False is the round-1 form (every statistic a standalone pass): same numbers, no records."""
    be = emu_backend
    be.fused_stats = False
    try:
        assert C.case_conv_moments(be, 1, 8, 32, (4, 4, 8), expect_fused=False) < 2e-5
        r = C.case_gn_bwd_fused(be, 1, 32, 32, (4, 4, 8), expect_fused=False)
        assert all(v < 2e-5 for v in r.values()), r
    finally:
        be.fused_stats = True


@pytest.mark.parametrize("lo,tgt", [((3, 4, 5), (6, 8, 10)), ((4, 3, 5), (7, 5, 9)), ((2, 2, 2), (5, 4, 4))])
def test_upsample(emu_backend, lo, tgt):
    assert ok(C.case_upsample(emu_backend, 2, 8, lo, tgt))


@pytest.mark.parametrize("ratio", [10.0, 30.0])
def test_groupnorm_large_mean(emu_backend, ratio):
    # |mean| >> std (ConvTranspose bias + zero-padded planes produce such groups): shifted-sum statistics and the centred
    # backward must stay at fp32 roundoff; E[x^2]-E[x]^2 loses every digit at ratio 30
    r = C.case_gn(emu_backend, 1, 64, (6, 6, 6), 8, offset=1.7 * ratio)
    assert all(v < 2e-5 for v in r.values()), r


@pytest.mark.parametrize("kw", [dict(n=2, cin=32, cout=32, dhw=(3, 4, 5)), dict(n=1, cin=64, cout=36, dhw=(2, 2, 3), norm=False, yld=72)])
def test_transposed_conv_k2s2(emu_backend, kw):
    assert ok(C.case_tconv2(emu_backend, **kw))


def test_conv_over_concat_with_per_channel_prologue(emu_backend):
    assert ok(C.case_conv_cat_slope(emu_backend, 1, 32, 32, 32, (5, 6, 7)))
    assert ok(C.case_conv_cat_slope(emu_backend, 1, 32, 32, 64, (5, 6, 7), stride=2))


def test_proj(emu_backend):
    assert ok(C.case_proj(emu_backend, 2, 32, 3, (5, 6, 7)))
    assert ok(C.case_proj(emu_backend, 1, 64, 3, (9, 6, 7), bias=True))
    assert ok(C.case_proj(emu_backend, 1, 32, 3, (5, 6, 7), bias=True, norm=True))


def test_dice(emu_backend):
    assert ok(C.case_dice(emu_backend, 2, 3, (12, 12, 12)))
    assert ok(C.case_dice(emu_backend, 2, 3, (12, 12, 12), batch=True, squared=True, u8=False))


def test_adam(emu_backend):
    assert C.case_adam(emu_backend, 1003) < 1e-5
    assert C.case_adam(emu_backend, 4096, wd=0.01) < 1e-5


def test_layout(emu_backend):
    assert C.case_layout(emu_backend, 2, 4, (5, 6, 7)) == 0.0


# ---- split-bf16 matrix path of the 3x3x3 stride-1 convs (csrc/conv3d_bf16.hip): fp32 in/out, products on bf16 MFMA ----
BF16_TOL = {"bf16x3": 1e-4, "bf16x6": 5e-6, "bf16": 3e-2, "fp16": 4e-3}      # fp16: MI355_PREC_F16, 11 significand bits


@pytest.fixture(params=["bf16x3", "bf16x6", "bf16", "fp16"])
def prec_backend(emu_backend, request):
    emu_backend.set_precision(request.param)
    yield emu_backend, BF16_TOL[request.param]
    emu_backend.set_precision("fp32")


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=32, dhw=(3, 5, 19), norm=True, residual=True, chscale=True),
    dict(n=1, cin=8, cout=64, dhw=(4, 4, 16), norm=True, yld=128, yc0=32),
    dict(n=1, cin=4, cout=32, dhw=(3, 4, 17), bias=True),
    dict(n=1, cin=48, cout=40, dhw=(2, 6, 9), norm=True, slope=0.01),
])
def test_conv_fwd_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    assert C.case_conv_fwd(be, **kw) < tol


# the 4-input-channel first layer on the bf16 pipe (conv3d_c4_fwd_bf16: K = fused (tap, ci) index, split operands, z-walking workgroups)
@pytest.mark.parametrize("kw", [
    dict(n=1, cin=4, cout=32, dhw=(9, 9, 9), norm=True),                                     # ragged tiles in every axis, 3 z tiles per column
    dict(n=2, cin=4, cout=64, dhw=(5, 6, 17), norm=True, residual=True, chscale=True),       # two co tiles (DynUNet width), two samples
    dict(n=1, cin=4, cout=48, dhw=(3, 4, 5), bias=True, yld=64, yc0=16),                     # partial co tile, concat slice
])
def test_first_layer_fwd_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    assert C.case_conv_fwd(be, **kw) < tol


def test_first_layer_moments_bf16_paths(prec_backend):
    be, tol = prec_backend
    assert C.case_conv_moments(be, 2, 4, 32, (9, 8, 10), ytol=tol, strict_vs_oracle=False) < 2e-5


@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=64, dhw=(3, 4, 18)), dict(n=2, cin=64, cout=32, dhw=(4, 4, 16))])
def test_conv_dgrad_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    assert C.case_conv_dgrad(be, **kw) < tol


@pytest.mark.parametrize("kw", [
    dict(n=2, cin=32, cout=32, dhw=(3, 5, 18), norm=True),
    dict(n=1, cin=4, cout=64, dhw=(4, 4, 16)),
    dict(n=1, cin=40, cout=96, dhw=(5, 3, 7), norm=True, slope=0.01),
    dict(n=2, cin=32, cout=32, dhw=(5, 8, 32), norm=True),      # 40 tiles in 5 splits: workgroups start mid-column and cross columns
    dict(n=1, cin=32, cout=32, dhw=(13, 6, 20), norm=True),     # long columns, ragged y / x tiles
])
def test_conv_wgrad_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    assert C.case_conv_wgrad(be, **kw) < tol


def test_conv_cat_slope_bf16_paths(prec_backend):
    # per-channel slope table (MONAI DynUNet concat) through the 16-bit kernels; forward (tile form) and weight gradient
    be, tol = prec_backend
    r = C.case_conv_cat_slope(be, 1, 32, 32, 32, (9, 6, 17))
    assert r["fwd"] < tol and r["wgrad"] < tol, r


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=32, dhw=(3, 5, 19), residual=True, chscale=True),     # ragged 2x4x16 tiles
    dict(n=1, cin=16, cout=64, dhw=(4, 4, 16), yld=128, yc0=32),                 # 2x2 wave grid, concat slice
    dict(n=2, cin=16, cout=40, dhw=(2, 6, 9), groups_out=40),                    # partial channel tile, InstanceNorm groups
])
def test_conv_epilogue_moments_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    assert C.case_conv_moments(be, ytol=tol, strict_vs_oracle=False, **kw) < 2e-5


@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=32, dhw=(3, 5, 19)), dict(n=2, cin=64, cout=16, dhw=(4, 4, 16), slope=0.01)])
def test_norm_backward_sums_from_dgrad_epilogue_bf16_paths(prec_backend, kw):
    be, tol = prec_backend
    r = C.case_gn_bwd_fused(be, compare_unfused=True, **kw)
    assert all(v < 2e-5 for v in r.values()), r


# ---- plane-ring forms of the 16-bit forward / dgrad kernel (csrc/conv3d_bf16.hip). MI355_BF16_FORM=zring: conv3d_k3_lp_zring2 (round 4:
#      input-plane-major, three output planes in flight, 17..64 input channels split over wave pairs above 32, output channels in tiles
#      of 32, >= 4 planes per z range); =zring1: conv3d_k3_lp_zring (round 3: 17..32 -> 32 channels). H % 8 == 0, W % 16 == 0. Shapes a
#      form does not take fall back to the tile kernel (the tolerance holds either way; the routing tests below pin which kernel ran). ----
@pytest.fixture(params=[("bf16", "", "zring"), ("fp16", "", "zring"), ("bf16", "2", "zring"), ("bf16", "5", "zring"), ("bf16", "", "zring1"), ("bf16", "2", "zring1")],
                ids=["bf16", "fp16", "bf16-two-z-ranges", "bf16-five-z-ranges", "v1-bf16", "v1-bf16-two-z-ranges"])
def zring_backend(emu_backend, request, monkeypatch):
    monkeypatch.setenv("MI355_BF16_FORM", request.param[2])
    monkeypatch.setenv("MI355_BF16_ZSPLITS", request.param[1])
    emu_backend.set_precision(request.param[0])
    yield emu_backend, BF16_TOL[request.param[0]]
    emu_backend.set_precision("fp32")


@pytest.mark.parametrize("kw", [
    dict(n=1, cin=32, cout=32, dhw=(5, 8, 16), norm=True, residual=True, chscale=True),      # one column, every epilogue operand
    dict(n=2, cin=24, cout=32, dhw=(6, 16, 32), bias=True),                                  # 8 columns, padded input channels
    dict(n=1, cin=32, cout=32, dhw=(1, 8, 16)),                                              # a single plane (zring2: tile kernel)
    dict(n=1, cin=20, cout=32, dhw=(9, 8, 32), norm=True, slope=0.01, yld=64, yc0=32),        # concat slice, leaky slope, 9 planes (two ranges: 5 + 4)
    dict(n=1, cin=32, cout=32, dhw=(4, 8, 16), norm=True),                                   # the shortest range: head + tail steps only
    dict(n=1, cin=32, cout=32, dhw=(22, 8, 16), residual=True),                              # every rotation of the three accumulator sets, 5 ranges of 5 / 4 / 4 ...
    dict(n=1, cin=64, cout=32, dhw=(6, 8, 16), norm=True, residual=True, chscale=True),      # channel split over wave pairs (KS = 2): the partial-tile exchange
    dict(n=2, cin=64, cout=64, dhw=(8, 16, 16), norm=True, bias=True),                       # two output-channel tiles per column
    dict(n=1, cin=48, cout=96, dhw=(7, 8, 32), slope=0.01, norm=True, yld=128, yc0=32),       # padded second channel slice (48 of 64), three channel tiles
    dict(n=1, cin=32, cout=64, dhw=(5, 16, 16), residual=True),                              # KS = 1 with two channel tiles
])
def test_conv_fwd_zring_form(zring_backend, kw):
    be, tol = zring_backend
    assert C.case_conv_fwd(be, **kw) < tol


@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=32, dhw=(6, 8, 16)), dict(n=1, cin=32, cout=64, dhw=(6, 8, 16)), dict(n=2, cin=64, cout=64, dhw=(5, 8, 32))])
def test_conv_dgrad_zring_form(zring_backend, kw):
    be, tol = zring_backend
    assert C.case_conv_dgrad(be, **kw) < tol


@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=32, dhw=(5, 8, 16), residual=True, chscale=True), dict(n=2, cin=32, cout=32, dhw=(4, 16, 16)),
                                dict(n=1, cin=64, cout=64, dhw=(9, 8, 16), residual=True)])
def test_conv_epilogue_moments_zring_form(zring_backend, kw):
    be, tol = zring_backend
    assert C.case_conv_moments(be, ytol=tol, strict_vs_oracle=False, **kw) < 2e-5


@pytest.mark.parametrize("kw", [dict(n=1, cin=32, cout=32, dhw=(5, 8, 16)), dict(n=1, cin=64, cout=32, dhw=(6, 8, 16)), dict(n=1, cin=32, cout=64, dhw=(4, 8, 16))])
def test_norm_backward_sums_zring_form(zring_backend, kw):
    be, tol = zring_backend
    # zring2 has no norm-backward epilogue (csrc/conv3d_bf16.hip: plan_lp_zring): a dgrad with 17..32 input and 32 output channels (the
    # forward layer's cout / cin) takes the round-3 kernel and fuses the sums; any other shape answers the statistics query with 0 -- the
    # sums take their own pass while the conv itself still runs on the plane-ring kernel
    fused = os.environ["MI355_BF16_FORM"] != "zring" or (16 < kw["cout"] <= 32 and kw["cin"] == 32)
    r = C.case_gn_bwd_fused(be, compare_unfused=True, expect_fused=fused, **kw)
    assert all(v < 2e-5 for v in r.values()), r


def _stats_blocks(be, n, dd, h, w, cin, cout):
    import ctypes
    d = C.ops._lib.MiConvDesc(); d.kd, d.stride, d.pad, d.precision = 3, 1, 1, be.precision
    x = be.empty_act(n, dd, h, w, cin); y = be.empty_act(n, dd, h, w, cout)
    xd, yd = x.desc(), y.desc()
    d.out_d, d.out_h, d.out_w = dd, h, w
    return be.lib.mi355_conv3d_stats_blocks(ctypes.byref(xd), ctypes.byref(yd), ctypes.byref(d))


def test_zring_form_default_routing(emu_backend, monkeypatch):
    """Without the switch (default `auto`) an eligible 16-bit layer with enough whole-CU workgroups takes the plane-ring kernel -- 64
    columns x 2 z ranges of 4 planes -- and a layer with fewer workgroups, or fewer than 4 planes, keeps the tile kernel. The kernel
    that ran is identified by its statistics records: eight (one per wave and half-wave) per (z range, column) against one per
    2 x 4 x 16 tile."""
    monkeypatch.delenv("MI355_BF16_FORM", raising=False)
    monkeypatch.delenv("MI355_BF16_ZSPLITS", raising=False)
    be = emu_backend
    be.set_precision("bf16")
    try:
        assert C.case_conv_fwd(be, n=2, cin=32, cout=32, dhw=(8, 32, 128), norm=True, residual=True) < BF16_TOL["bf16"]
        for (n, dd, h, w, cin, cout), want in (((2, 8, 32, 128, 32, 32), 2 * 4 * 8 * 8),    # plane ring: 2 z ranges x 32 columns per sample x 8 (wave, half-wave)
                                               ((2, 8, 32, 128, 64, 64), 2 * 4 * 8 * 8),    # the same with two channel tiles (records are per channel)
                                               ((1, 4, 8, 16, 32, 32), 2 * 2 * 1),          # 1 column: 2 x 4 x 16 tiles of the tile kernel
                                               ((2, 3, 32, 128, 32, 32), 2 * 8 * 8)):       # 3 planes: tile kernel
            assert _stats_blocks(be, n, dd, h, w, cin, cout) == want, (n, dd, h, w, cin, cout)
    finally:
        be.set_precision("fp32")


def test_zring_form_is_what_ran(emu_backend, monkeypatch):
    """The switch routes an eligible call to the plane-ring kernel (its statistics records are per (z range, column), not per tile):
    guards the tests above against silently exercising the tile kernel. 64-channel layers: zring2 only."""
    be = emu_backend
    be.set_precision("bf16")
    try:
        monkeypatch.setenv("MI355_BF16_FORM", "tile")
        tile_blocks = _stats_blocks(be, 1, 8, 8, 16, 32, 32)
        tile_blocks64 = _stats_blocks(be, 1, 8, 8, 16, 64, 64)
        monkeypatch.setenv("MI355_BF16_ZSPLITS", "2")
        monkeypatch.setenv("MI355_BF16_FORM", "zring")
        assert _stats_blocks(be, 1, 8, 8, 16, 32, 32) == 2 * 8 != tile_blocks   # zring2: 8 records per (z range, column)
        assert _stats_blocks(be, 1, 8, 8, 16, 64, 64) == 2 * 8 != tile_blocks64
        assert _stats_blocks(be, 1, 9, 8, 16, 48, 96) == 2 * 8                  # 5 + 4 planes
        assert _stats_blocks(be, 1, 7, 8, 16, 32, 32) == 1 * 8                  # 7 planes cannot make two ranges of >= 4
        monkeypatch.setenv("MI355_BF16_FORM", "zring1")
        assert _stats_blocks(be, 1, 8, 8, 16, 32, 32) == 2
        assert _stats_blocks(be, 1, 8, 8, 16, 64, 64) == tile_blocks64          # the round-3 kernel does not take 64 channels
    finally:
        be.set_precision("fp32")


# ---- plane-ring wgrad on tall planes (>= 32 rows: many columns per workgroup, z chunks) ----
@pytest.mark.parametrize("kw", [
    dict(n=2, cin=32, cout=32, dhw=(3, 34, 9), norm=True),            # ragged columns in y (34 = 8 x 4 + 2) and x, two samples
    dict(n=1, cin=40, cout=24, dhw=(2, 33, 17)),                      # partial channel tiles, plain input
    dict(n=1, cin=32, cout=32, dhw=(32, 32, 8), norm=True),           # columns cut into z chunks
])
def test_conv_wgrad_ring_tall_planes(emu_backend, kw):
    assert C.case_conv_wgrad(emu_backend, **kw) < TOL


# ---- the interior-tile epilogue and the general one must store the same bits ----
@pytest.mark.parametrize("prec,kd,cin,cout,dhw", [
    ("fp32", 3, 32, 32, (4, 8, 16)), ("fp32", 3, 16, 64, (8, 8, 8)), ("fp32", 3, 64, 32, (4, 4, 8)),
    ("bf16", 3, 32, 32, (4, 8, 32)), ("bf16x3", 3, 16, 64, (4, 4, 16)), ("fp16", 3, 32, 32, (4, 4, 16)),
])
def test_interior_and_general_epilogues_store_identical_values(emu_backend, prec, kd, cin, cout, dhw):
    """A layer whose extents are tile multiples takes the interior-tile epilogue (one base pointer per lane, wave-uniform offsets) in
    every workgroup; writing the same result into a destination that is one voxel larger per axis, shifted by (1, 1, 1), forces the
    general epilogue (per-value lane map, bound checks, window test). Same accumulators, same bias / residual / dropout-scale
    arithmetic: the stored values must be bitwise equal."""
    be = emu_backend
    g = torch.Generator().manual_seed(21)
    d, h, w = dhw
    n = 2
    x = torch.randn(n, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, kd, kd, kd, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, d, h, w, generator=g)
    cs = (torch.rand(n, cout, generator=g) > 0.3).float() * 1.25
    be.set_precision(prec)
    try:
        xa, ra = C.to_act(be, x), C.to_act(be, res)
        wp = be.pack_weight(wt, 0)
        y1 = C.to_act(be, torch.zeros(n, cout, d, h, w))
        be.conv_fwd(xa, wp, y1, kd, 1, bias=b, residual=ra, chscale=cs)
        y2 = C.to_act(be, torch.full((n, cout, d + 1, h + 1, w + 1), 7.0))
        be.conv_fwd(xa, wp, y2, kd, 1, bias=b, residual=ra, chscale=cs, off=(1, 1, 1), out_dhw=(d, h, w))
    finally:
        be.set_precision("fp32")
    a, c2 = C.from_act(y1), C.from_act(y2)
    assert torch.equal(a, c2[:, :, 1:, 1:, 1:])
    assert bool((c2[:, :, 0] == 7.0).all()) and bool((c2[:, :, :, 0] == 7.0).all()) and bool((c2[..., 0] == 7.0).all())


def test_batched_repack_equals_single_packs(emu_backend):
    """mi355_pack_weights_batch (one launch for every pack of a step) writes exactly what the single-weight pack kernels write: fp32
    packs in the four modes, Winograd packs in both, ragged channel counts."""
    be = emu_backend
    g = torch.Generator().manual_seed(11)
    ws = [(torch.randn(40, 12, 3, 3, 3, generator=g), 0), (torch.randn(40, 12, 3, 3, 3, generator=g), 1), (torch.randn(8, 24, 1, 1, 1, generator=g), 0),
          (torch.randn(16, 8, 3, 3, 3, generator=g), 2), (torch.randn(16, 8, 3, 3, 3, generator=g), 3)]
    packs = [be.pack_weight(w, m) for w, m in ws]
    for pw in packs:
        pw.f32()
    packs[0].wino(); packs[1].wino()
    for w, _ in ws:                                   # "optimizer step": the weights change in place
        w.mul_(-0.5).add_(0.25)
    assert be.repack_batch(packs) == 7
    fresh = [be.pack_weight(w, m) for w, m in ws]
    for a, b in zip(packs, fresh):
        assert torch.equal(a._f32, b.f32())
    assert torch.equal(packs[0]._wino, fresh[0].wino()) and torch.equal(packs[1]._wino, fresh[1].wino())
    table = be.last_pack_table
    assert be.repack_batch(packs) == 7 and be.last_pack_table is table    # the device table is reused
    own = {}
    assert be.repack_batch(packs, own) == 7 and len(own) == 1 and be.last_pack_table is not table      # a caller's own cache gets its own table


@pytest.mark.parametrize("prec", ["bf16", "fp16", "bf16x3", "bf16x6"])
def test_batched_repack_refreshes_the_16bit_packs(emu_backend, prec):
    """The 16-bit operand packs of a step ride in the same launch (kind MI355_PACK_LP + precision): bit-identical to
    mi355_pack_conv_weight_bf16 in both modes, ragged channels; a pack of a precision mode that is no longer in use is dropped."""
    be = emu_backend
    g = torch.Generator().manual_seed(12)
    pnum = C.ops.PRECISIONS[prec]
    ws = [(torch.randn(40, 20, 3, 3, 3, generator=g), 0), (torch.randn(40, 20, 3, 3, 3, generator=g), 1), (torch.randn(32, 64, 3, 3, 3, generator=g), 0)]
    packs = [be.pack_weight(w, m) for w, m in ws]
    for pw in packs:
        pw.f32()
        pw.bf16(pnum)
    packs[2].bf16(C.ops.PRECISIONS["bf16x3" if prec != "bf16x3" else "bf16"])      # an older pack of another mode ...
    packs[2].bf16(pnum)                                                            # ... the mode in use is the last one asked for
    for w, _ in ws:
        w.mul_(0.75).add_(-0.125)
    be.set_precision(prec)                            # the mode the next forward runs in decides which 16-bit pack is refreshed
    try:
        assert be.repack_batch(packs) == 6
        assert set(packs[2]._bf16) == {pnum}
        for pw, (w, m) in zip(packs, ws):
            fresh = be.pack_weight(w, m)
            assert torch.equal(pw._f32, fresh.f32()) and torch.equal(pw._bf16[pnum], fresh.bf16(pnum))
    finally:
        be.set_precision("fp32")
    # back in fp32 nobody reads the 16-bit packs: they are dropped (3 fp32 tasks left), not rewritten every step
    assert be.repack_batch(packs) == 3 and all(not pw._bf16 for pw in packs)


@pytest.mark.parametrize("conv_precision", [None, "bf16"])
def test_training_step_repacks_in_one_launch(emu_backend, conv_precision):
    """(conv_precision "bf16": a network with its own mode on a backend that is in fp32 between forwards -- the 16-bit packs of THAT
    mode ride in the batch, they are not dropped and rebuilt by single launches every step.)
    From the second step on, the packs of a training step are refreshed by ONE mi355_pack_weights_batch launch at the start of the
    forward (engine.py: _repack_stale) and no single-weight pack kernel runs; the step computes what a model with freshly built packs
    computes from the same weights."""
    import importlib
    unet = importlib.import_module("3dunetcnn_amd.unet")
    losses = importlib.import_module("3dunetcnn_amd.losses")
    optim = importlib.import_module("3dunetcnn_amd.optim")
    from oracle import unet3d_ref as R
    be = emu_backend
    torch.manual_seed(5)
    kw = dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1])
    m = unet.HipUNet3D(**kw).eval()
    m._be = be
    m.conv_precision = conv_precision
    crit, opt = losses.HipDiceLoss(sigmoid=True), optim.HipAdam(m.parameters(), lr=1e-2)
    crit._be = opt._be = be
    x, y = R.synthetic_case(1, 4, (16, 16, 16), 3)
    counts = {"single": 0, "batch": 0}
    orig = {n: getattr(be.lib, n) for n in ("mi355_pack_conv_weight", "mi355_wino_pack_weight", "mi355_pack_weights_batch",
                                            "mi355_pack_conv_weight_bf16")}

    def counted(name, key):
        def f(*a):
            counts[key] += 1
            return orig[name](*a)
        return f
    be.lib.mi355_pack_conv_weight = counted("mi355_pack_conv_weight", "single")
    be.lib.mi355_wino_pack_weight = counted("mi355_wino_pack_weight", "single")
    be.lib.mi355_pack_weights_batch = counted("mi355_pack_weights_batch", "batch")
    be.lib.mi355_pack_conv_weight_bf16 = counted("mi355_pack_conv_weight_bf16", "single")
    try:
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            crit(m(x), y).backward()
            opt.step()
        first = dict(counts)
        assert first["single"] > 10 and first["batch"] == 1          # step 1 packs lazily, step 2 starts with the batch
        opt.zero_grad(set_to_none=True)
        out = m(x)
        crit(out, y).backward()
        assert counts["batch"] == 2 and counts["single"] == first["single"], counts
    finally:
        for n, f in orig.items():
            setattr(be.lib, n, f)
    m2 = unet.HipUNet3D(**kw).eval()                   # the same weights through freshly built packs
    m2._be = be
    m2.conv_precision = conv_precision
    m2.load_state_dict(m.state_dict())
    out2 = m2(x)
    crit(out2, y).backward()
    assert torch.equal(out.detach(), out2.detach())
    for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(p.grad, q.grad), k



@pytest.mark.parametrize("kw", [
    dict(n=1, dhw=(3, 8, 16)),                                   # one column tile, one z chunk
    dict(n=2, dhw=(5, 9, 19), groups=2, slope=0.01),             # ragged tiles, leaky slope, two channels per group
    dict(n=1, dhw=(17, 8, 16), groups=1),                        # two z chunks: the halo planes of a chunk come from its neighbour
    dict(n=1, dhw=(2, 3, 5), xld=8, dyld=64),                    # image smaller than the tile; views of wider buffers
])
def test_first_layer_fused_backward(emu_backend, kw):
    r = C.case_c4_bwd(emu_backend, **kw)
    assert all(v < 1e-4 for v in r.values()), r
