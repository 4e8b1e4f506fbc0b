"""16-bit activation storage (mi355_act.dtype = MI355_ACT_BF16 / MI355_ACT_F16) at op level on the CPU emulator build of the kernel sources (index logic,
type dispatch, rounding points); tests/act_storage_cases.py states the property. GPU twins: tests/test_act_storage_gpu.py."""
import importlib

import pytest
import torch

import act_storage_cases as S

ops = importlib.import_module("3dunetcnn_amd.ops")
TOL = S.TOL


@pytest.fixture
def bf16_backend(emu_backend):
    saved = emu_backend.precision
    emu_backend.set_precision("bf16")
    yield emu_backend
    emu_backend.precision = saved


def _all_below(errs, tol=TOL, **special):
    bad = {k: v for k, v in errs.items() if not isinstance(v, bool) and v > special.get(k, tol)}
    assert not bad, (bad, errs)


def test_cast(emu_backend):
    S.case_cast(emu_backend)


def test_pointwise_ops(emu_backend):
    _all_below(S.case_pointwise(emu_backend))


@pytest.mark.parametrize("kw", [dict(), dict(n=1, c=4, dhw=(4, 4, 6), groups=4), dict(c=24, groups=24, slope=0.01)])
def test_norm_statistics_and_backward(emu_backend, kw):
    _all_below(S.case_norm(emu_backend, **kw), dgamma=1e-5, dbeta=1e-5)


def test_projection(emu_backend):
    _all_below(S.case_proj(emu_backend), dw=1e-5)


@pytest.mark.parametrize("kw", [dict(), dict(cin=32, cout=64), dict(cin=64, cout=64, residual=True), dict(cin=32, cout=64, residual=True, dhw=(3, 3, 5)),
                                dict(cin=128, cout=64)])      # the first four: conv3d_k1_stream_bf16 (ragged last chunk; 90 voxels); the last: the template
def test_conv_1x1x1(emu_backend, kw):
    _all_below(S.case_conv_k1(emu_backend, **kw))


@pytest.mark.parametrize("kw", [dict(), dict(cin=64, cout=32, moments=False)])
def test_conv_stride2(emu_backend, kw):
    _all_below(S.case_conv_s2(emu_backend, **kw), moments=2e-5)


@pytest.mark.parametrize("kw", [dict(), dict(window=True), dict(cin=32, cout=64), dict(cin=32, cout=32)])
def test_conv_zero_insert(emu_backend, kw):
    _all_below(S.case_conv_zero_insert(emu_backend, **kw))


@pytest.mark.parametrize("kw", [
    dict(cin=32, cout=32, dhw=(4, 4, 16)),                                             # plain input: staged without a conversion
    dict(cin=32, cout=32, dhw=(5, 6, 18), norm=True, moments=True),                    # ragged tiles, norm prologue, moment records
    dict(cin=64, cout=32, dhw=(4, 8, 16), norm=True, residual=True, drop=True, moments=True, n=2),
    dict(cin=32, cout=64, dhw=(4, 5, 17), gnb=True, mode=1),                           # dgrad with the norm-backward sums, ragged
    dict(cin=40, cout=24, dhw=(3, 4, 9), norm=True),                                   # channel counts that are not tile multiples
])
def test_conv_3x3x3_on_16bit_operands(bf16_backend, kw):
    r = S.case_conv_k3_tile(bf16_backend, **kw)
    if kw.get("gnb"):
        assert r["gnb_fused"]
    _all_below(r, moments=2e-5, gnb=1e-5)


@pytest.mark.parametrize("wide,kw", [
    ("big", dict(cin=32, cout=128, dhw=(4, 4, 16), norm=True, residual=True, moments=True)),    # 4 x 4 x 16 tile, 4 x 2 accumulator tiles per wave
    ("small", dict(cin=16, cout=128, dhw=(3, 5, 17), mode=1)),                                  # 2 x 4 x 16 tile, 2 x 2 tiles per wave, ragged
    ("big", dict(cin=16, cout=128, dhw=(4, 4, 16), gnb=True, mode=1)),                          # norm-backward sums on whole tiles: the wide form
    ("big", dict(cin=16, cout=128, dhw=(4, 5, 16), gnb=True, mode=1)),                          # ... ragged: the 64-channel form
    ("small", dict(cin=16, cout=128, dhw=(2, 4, 16), gnb=True, mode=1)),                        # ... the small tile keeps 64 channels with these sums
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
    ("zring", dict(cin=32, cout=32, dhw=(6, 8, 16), norm=True, moments=True)),
    ("zring", dict(cin=64, cout=32, dhw=(5, 8, 32), norm=True, residual=True, drop=True, moments=True, n=2)),
    ("zring", dict(cin=32, cout=64, dhw=(4, 16, 16), mode=1)),                                  # plain input (a dgrad): staged as loaded
    ("zring", dict(cin=64, cout=64, dhw=(9, 8, 16), norm=True, residual=True)),
    ("zring1", dict(cin=32, cout=32, dhw=(5, 8, 16), gnb=True, mode=1)),                        # the round-3 ring: norm-backward sums
    ("zring1", dict(cin=24, cout=32, dhw=(4, 8, 16), norm=True, moments=True, residual=True)),
])
def test_conv_3x3x3_plane_ring_forms(bf16_backend, monkeypatch, form, kw):
    """The plane-ring kernels (csrc/conv3d_bf16_zring.hip) on 16-bit tensors: MI355_BF16_FORM=zring forces conv3d_k3_lp_zring2 on every
    eligible shape, zring1 the round-3 kernel."""
    monkeypatch.setenv("MI355_BF16_FORM", form)
    r = S.case_conv_k3_tile(bf16_backend, **kw)
    if kw.get("gnb"):
        assert r["gnb_fused"]
    _all_below(r, moments=2e-5, gnb=1e-5)


def test_first_layer(bf16_backend):
    _all_below(S.case_first_layer(bf16_backend), moments=2e-5, wgrad=1e-5, c4bwd_vs_wgrad=1e-5)


@pytest.mark.parametrize("kw", [
    dict(kd=1, stride=1, cin=64, cout=32, dhw=(4, 5, 7)),                              # conv3d_wgrad_k1_lp_tr: 2 x 1 tiles, 140 voxels (ragged last chunk)
    dict(kd=1, stride=1, cin=32, cout=64, dhw=(5, 5, 7), n=2),                         # ... 1 x 2 tiles
    dict(kd=1, stride=1, cin=64, cout=128, dhw=(9, 9, 17), n=2),                       # ... 2 x 4 tiles, several workgroups
    dict(kd=1, stride=1, cin=128, cout=64, dhw=(3, 7, 11)),                            # ... 4 x 2 tiles
    dict(kd=1, stride=1, cin=64, cout=64, dhw=(4, 5, 7)),                              # (2 x 2 tiles: the generic kernel)
    dict(kd=3, stride=2, cin=32, cout=32, dhw=(9, 8, 11), n=2),
    dict(kd=3, stride=1, cin=32, cout=32, dhw=(5, 6, 18), norm=True),                  # 16-bit-operand weight gradient
    dict(kd=3, stride=1, cin=32, cout=96, dhw=(3, 4, 17), norm=True),                  # ... its 64-channel workgroup form
    dict(kd=3, stride=1, cin=32, cout=32, dhw=(5, 9, 18), norm=True),                  # conv3d_wgrad_lp_tr (H >= 8, W >= 16): ragged tiles in y and x
    dict(kd=3, stride=1, cin=64, cout=32, dhw=(17, 8, 16), n=2),                       # ... plain input, two (ci, co) pairs, two z chunks per column
    dict(kd=3, stride=1, cin=32, cout=64, dhw=(3, 11, 35), norm=True, n=2),            # ... fewer planes than the ring holds
])
def test_weight_gradients(bf16_backend, kw):
    _all_below(S.case_wgrad(bf16_backend, **kw), dw=1e-5)


def test_mixed_storage_types_are_refused(emu_backend):
    """One storage type per call (except the first-layer kernels): the library answers MI355_STATUS_EUNSUPPORTED, never reinterprets."""
    be = emu_backend
    a = be.empty_act(1, 4, 4, 8, 32)
    b = be.empty_act(1, 4, 4, 8, 32, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="unsupported"):
        be.add(a, b, a)
    w = be.pack_weight(torch.randn(32, 32, 1, 1, 1), 0)
    with pytest.raises(RuntimeError, match="unsupported"):
        be.conv_fwd(a, w, b, 1)
    w3 = be.pack_weight(torch.randn(32, 32, 3, 3, 3), 0)
    y = be.empty_act(1, 4, 4, 8, 32, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="unsupported"):
        be.conv_fwd(b, w3, y, 3)          # fp32 arithmetic (the backend's default precision) on 16-bit tensors: no such 3x3x3 kernel


# ---- IEEE fp16 storage (MI355_ACT_F16: the tensors of the reference's own amp mode, train/train.py:33-37) with fp16 operands: the same
# ---- property on the other 16-bit type. A subset of the cases above per kernel family (every typed kernel has an f16_t instantiation).
@pytest.fixture
def fp16_backend(emu_backend):
    saved = emu_backend.precision
    emu_backend.set_precision("fp16")
    with S.storage_type(torch.float16):
        yield emu_backend
    emu_backend.precision = saved


def test_fp16_cast_pointwise_norm_projection(fp16_backend):
    S.case_cast(fp16_backend)
    _all_below(S.case_pointwise(fp16_backend))
    _all_below(S.case_norm(fp16_backend), dgamma=1e-5, dbeta=1e-5)
    _all_below(S.case_norm(fp16_backend, c=24, groups=24, slope=0.01), dgamma=1e-5, dbeta=1e-5)
    _all_below(S.case_proj(fp16_backend), dw=1e-5)


def test_fp16_conv_1x1x1_stride2_zero_insert(fp16_backend):
    _all_below(S.case_conv_k1(fp16_backend))
    _all_below(S.case_conv_s2(fp16_backend), moments=2e-5)
    _all_below(S.case_conv_zero_insert(fp16_backend, window=True))


@pytest.mark.parametrize("kw", [
    dict(cin=32, cout=32, dhw=(4, 4, 16)),                                             # plain input: staged without a conversion
    dict(cin=64, cout=32, dhw=(4, 8, 16), norm=True, residual=True, drop=True, moments=True, n=2),
    dict(cin=32, cout=64, dhw=(4, 5, 17), gnb=True, mode=1),                           # dgrad with the norm-backward sums, ragged
])
def test_fp16_conv_3x3x3_on_16bit_operands(fp16_backend, kw):
    r = S.case_conv_k3_tile(fp16_backend, **kw)
    if kw.get("gnb"):
        assert r["gnb_fused"]
    _all_below(r, moments=2e-5, gnb=1e-5)


@pytest.mark.parametrize("form,kw", [
    ("zring", dict(cin=32, cout=64, dhw=(4, 16, 16), mode=1)),                                  # plain input (a dgrad): staged as loaded
    ("zring", dict(cin=64, cout=32, dhw=(5, 8, 32), norm=True, residual=True, drop=True, moments=True, n=2)),
    ("zring1", dict(cin=32, cout=32, dhw=(5, 8, 16), gnb=True, mode=1)),                        # the round-3 ring: norm-backward sums
])
def test_fp16_conv_3x3x3_plane_ring_forms(fp16_backend, monkeypatch, form, kw):
    monkeypatch.setenv("MI355_BF16_FORM", form)
    r = S.case_conv_k3_tile(fp16_backend, **kw)
    if kw.get("gnb"):
        assert r["gnb_fused"]
    _all_below(r, moments=2e-5, gnb=1e-5)


def test_fp16_first_layer_and_weight_gradients(fp16_backend):
    _all_below(S.case_first_layer(fp16_backend), moments=2e-5, wgrad=1e-5, c4bwd_vs_wgrad=1e-5)
    _all_below(S.case_wgrad(fp16_backend, kd=1, stride=1, cin=64, cout=32, dhw=(4, 5, 7)), dw=1e-5)
    _all_below(S.case_wgrad(fp16_backend, kd=3, stride=2, cin=32, cout=32, dhw=(9, 8, 11), n=2), dw=1e-5)
    _all_below(S.case_wgrad(fp16_backend, kd=3, stride=1, cin=32, cout=32, dhw=(5, 6, 18), norm=True), dw=1e-5)
    _all_below(S.case_wgrad(fp16_backend, kd=3, stride=1, cin=32, cout=96, dhw=(3, 4, 17), norm=True), dw=1e-5)
    _all_below(S.case_wgrad(fp16_backend, kd=3, stride=1, cin=32, cout=32, dhw=(4, 10, 19), norm=True), dw=1e-5)      # conv3d_wgrad_lp_tr


def test_fp16_storage_goes_with_fp16_operands_only(emu_backend):
    """A 16-bit tensor is stored in the type its convolutions round their operands to: fp16 tensors under bf16 arithmetic (and the reverse)
    are refused, never reinterpreted."""
    be = emu_backend
    saved = be.precision
    try:
        w3 = be.pack_weight(torch.randn(32, 32, 3, 3, 3), 0)
        for prec, dt in (("bf16", torch.float16), ("fp16", torch.bfloat16)):
            be.set_precision(prec)
            x = be.empty_act(1, 4, 4, 16, 32, dtype=dt); x.buf.zero_()
            y = be.empty_act(1, 4, 4, 16, 32, dtype=dt)
            with pytest.raises(RuntimeError, match="unsupported"):
                be.conv_fwd(x, w3, y, 3)
    finally:
        be.precision = saved


def test_randomized_shapes(bf16_backend, monkeypatch):
    """A fixed-seed sample of the developer fuzz over the storage cases: random channel counts (incl. non-multiples of the tiles), ragged
    extents, fusion combinations and kernel forms (160 cases of it ran clean when the storage landed)."""
    import random
    rng = random.Random(4)
    be = bf16_backend
    for _ in range(10):                                  # (the plane-ring forms have their own cases above: minutes each on the emulator)
        kind = rng.choice(["tile", "tile", "k1", "s2", "zi", "wgrad", "norm"])
        if kind in ("tile", "ring", "ring1"):
            monkeypatch.setenv("MI355_BF16_FORM", {"tile": "tile", "ring": "zring", "ring1": "zring1"}[kind])
            if kind == "tile":
                kw = dict(cin=rng.choice([8, 24, 32, 40, 64]), cout=rng.choice([8, 32, 48, 64]), dhw=(rng.randint(1, 5), rng.randint(1, 9), rng.randint(1, 19)))
            else:
                kw = dict(cin=rng.choice([20, 24, 32]) if kind == "ring1" else rng.choice([20, 32, 48, 64]), cout=32 if kind == "ring1" else rng.choice([32, 64]),
                          dhw=(rng.randint(4, 8), 8, 16 * rng.randint(1, 2)))
            kw.update(norm=rng.random() < 0.5, residual=rng.random() < 0.5, drop=rng.random() < 0.3, moments=rng.random() < 0.5, n=rng.choice([1, 2]))
            if kind != "ring" and rng.random() < 0.3:
                kw.update(gnb=True, mode=1, norm=False, moments=False)
            r = S.case_conv_k3_tile(be, **kw)
        elif kind == "k1":
            kw = dict(cin=rng.choice([8, 32, 128]), cout=rng.choice([8, 32, 64])); r = S.case_conv_k1(be, **kw)
        elif kind == "s2":
            kw = dict(cin=rng.choice([8, 32, 64]), cout=rng.choice([16, 64]), moments=rng.random() < 0.5); r = S.case_conv_s2(be, **kw)
        elif kind == "zi":
            kw = dict(cin=rng.choice([32, 64]), cout=rng.choice([16, 32]), window=rng.random() < 0.5); r = S.case_conv_zero_insert(be, **kw)
        elif kind == "wgrad":
            kd = rng.choice([1, 3, 3]); stride = 1 if kd == 1 else rng.choice([1, 2])
            kw = dict(kd=kd, stride=stride, cin=rng.choice([8, 32, 40]), cout=rng.choice([16, 32, 96]), dhw=(rng.randint(2, 6), rng.randint(2, 8), rng.randint(3, 19)),
                      norm=rng.random() < 0.5 and kd == 3 and stride == 1)
            r = S.case_wgrad(be, **kw)
        else:
            c = rng.choice([4, 8, 16, 24])
            kw = dict(n=rng.choice([1, 2]), c=c, dhw=(rng.randint(1, 4), rng.randint(1, 6), rng.randint(1, 9)), groups=rng.choice([g for g in (1, 2, 4, c) if c % g == 0]))
            r = S.case_norm(be, **kw)
        bad = {k: v for k, v in r.items() if not isinstance(v, bool) and v > {"moments": 3e-5, "gnb": 2e-5, "dw": 1e-5, "dgamma": 1e-5, "dbeta": 1e-5, "stats": 1e-5}.get(k, TOL)}
        assert not bad, (kind, kw, bad)
