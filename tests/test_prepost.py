"""Steps either side of the network: on-device one-hot encode, label-map decode (+ activation) and intensity
normalisation vs oracle/prepost_ref.py, which is itself pinned here against the reference's live functions and against the
known answers of the reference's own tests."""
import importlib
import sys
import types

import pytest
import torch

from oracle import prepost_ref as P, reference_shim

prepost = importlib.import_module("3dunetcnn_amd.prepost")


def _reference_one_hot_module():
    reference_shim.import_reference_unet()
    if "monai.data" not in sys.modules:
        md = types.ModuleType("monai.data")

        class MetaTensor(torch.Tensor):        # minimal stand-in: a Tensor subclass that survives torch ops and has `.meta`
            meta = {}

            @staticmethod
            def __new__(cls, x, meta=None, **kw):
                return torch.Tensor._make_subclass(cls, x)
        md.MetaTensor = MetaTensor
        sys.modules["monai.data"] = md
        sys.modules["monai"].data = md
    return importlib.import_module("unet3d.utils.one_hot")


def _label_volume():
    # the fixture of the reference's test_compile_one_hot_encoding (test/test_utils.py:103-128)
    flat = torch.zeros(100 * 100 * 100)
    flat[10:3000] = 5
    flat[4000:90000] = 22
    return flat.reshape(1, 1, 100, 100, 100)


def test_oracle_reproduces_the_reference_tests_known_answers():
    data = _label_volume()
    t5 = (data[0] == 5).to(torch.uint8)
    t22 = (data[0] == 22).to(torch.uint8)
    assert torch.equal(P.compile_one_hot_encoding(data, 1, labels=[5]), t5)
    assert torch.equal(P.compile_one_hot_encoding(data, 2, labels=[5, 22]), torch.cat([t5, t22]))
    assert torch.equal(P.compile_one_hot_encoding(data, 1, labels=[[5, 22]]), (t5 | t22))
    # test/test_segment.py:8-19
    lm = torch.zeros((10, 10, 10), dtype=torch.int16)
    lm[4], lm[5] = 3, 17
    g = (lm > 0).to(torch.int16)
    left, right = torch.zeros_like(g), torch.zeros_like(g)
    left[:5], right[5:] = g[:5], g[5:]
    assert torch.equal(P.convert_one_hot_to_label_map(torch.stack((left, right)), [3, 17]), lm)


@pytest.mark.skipif(not reference_shim.available(), reason="/root/reference only exists in the build container")
def test_oracle_matches_live_reference_functions():
    ref = _reference_one_hot_module()
    g = torch.Generator().manual_seed(0)
    lm = torch.randint(0, 5, (1, 1, 9, 8, 7), generator=g).float()
    lm_meta = sys.modules["monai.data"].MetaTensor(lm)
    for labels in ([1, 2, 4], [[1, 2], [4]], None):
        n = 3 if labels is None else len(labels)
        want = ref.compile_one_hot_encoding(lm_meta, n, labels=labels).as_subclass(torch.Tensor)
        assert torch.equal(want, P.compile_one_hot_encoding(lm, n, labels=labels))
    oh = torch.rand(3, 9, 8, 7, generator=g)
    for kw in (dict(label_hierarchy=True), dict(), dict(sum_then_threshold=True), dict(threshold=0.7)):
        assert torch.equal(ref.convert_one_hot_to_label_map(oh, [2, 1, 4], **kw), P.convert_one_hot_to_label_map(oh, [2, 1, 4], **kw))
    assert torch.equal(ref.convert_one_hot_to_label_map(oh, [[2, 1], [4]]), P.convert_one_hot_to_label_map(oh, [[2, 1], [4]]))


def _cases(dev, be, dhw):
    g = torch.Generator().manual_seed(1)
    lm = torch.randint(0, 6, (1, 1, *dhw), generator=g).float()
    lm[0, 0, 0, 0, :3] = torch.tensor([1.4, 2.5, 3.5])        # rounding: half to even
    for labels in ([2, 1, 4], [[2, 1, 4], [1, 4], [4]], None):
        n = 3 if labels is None else len(labels)
        got = prepost.compile_one_hot_encoding(lm.to(dev), n, labels=labels, _backend=be)
        assert torch.equal(got.cpu(), P.compile_one_hot_encoding(lm, n, labels=labels))
    logits = torch.randn(3, *dhw, generator=g) * 2
    for act, ref_p in (("sigmoid", torch.sigmoid(logits)), ("softmax", torch.softmax(logits, 0)), (None, logits)):
        for kw in (dict(label_hierarchy=True), dict(), dict(sum_then_threshold=True)):
            probs, lab = prepost.activate_and_decode(logits.to(dev), act, [2, 1, 4], 0.5, _backend=be, **kw)
            assert torch.allclose(probs.cpu(), ref_p, atol=1e-6)
            want = P.convert_one_hot_to_label_map(probs.cpu(), [2, 1, 4], 0.5, kw.get("sum_then_threshold", False), torch.int16,
                                                  kw.get("label_hierarchy", False))
            assert torch.equal(lab.cpu(), want)
    oh = torch.rand(3, *dhw, generator=g)
    got = prepost.convert_one_hot_to_label_map(oh.to(dev), [[2, 1], [4]], _backend=be)
    assert torch.equal(got.cpu(), P.convert_one_hot_to_label_map(oh, [[2, 1], [4]]))
    x = torch.randn(4, *dhw, generator=g) * torch.tensor([1.0, 30.0, 0.01, 5.0]).view(4, 1, 1, 1) + torch.tensor([0.0, 500.0, -3.0, 80.0]).view(4, 1, 1, 1)
    x[2] = 7.0                                                 # constant channel: std 0 -> divisor 1
    got = prepost.normalize_intensity(x.to(dev), _backend=be)
    assert torch.allclose(got.cpu(), P.normalize_intensity(x), atol=2e-5)


def _resample_cases(dev, be, dhw):
    """ResizeD / ResampleToMatch on device vs torch F.interpolate / F.grid_sample. Tolerance 1e-4 absolute on N(0,1) data: the
    source coordinate is an fp32 number of magnitude <= 150 (ulp 1.5e-5 at 128..256), so two correct evaluations differ by
    ulp(coordinate) x |neighbour difference|; nearest modes are bit-exact away from exact .5 / integer-boundary coordinates."""
    g = torch.Generator().manual_seed(3)
    img = torch.randn(3, *dhw, generator=g)
    lab = torch.randint(0, 5, (1, *dhw), generator=g).float()
    for size in ((dhw[0] * 2, dhw[1] * 2, dhw[2] * 2), (dhw[0] + 3, dhw[1] - 2, dhw[2] * 3 - 1), (5, 4, 3), dhw):
        got = prepost.resize(img.to(dev), size, "trilinear", _backend=be)
        assert got.shape == (3, *size)
        err = float((got.cpu() - P.resize_ref(img, size, "trilinear")).abs().max())
        assert err < 1e-4, (size, err)
        gotl = prepost.resize(lab.to(dev), size, "nearest", _backend=be)
        assert torch.equal(gotl.cpu(), P.resize_ref(lab, size, "nearest")), size
    # prediction grid (2 mm, shifted origin, one flipped axis) -> source grid (1 mm)
    a_src = torch.tensor([[2.0, 0, 0, -10.0], [0, -2.0, 0, 31.0], [0, 0, 2.0, 4.0], [0, 0, 0, 1.0]])
    a_dst = torch.tensor([[1.0, 0, 0, -11.3], [0, 1.0, 0, 1.5], [0, 0, 1.0, 3.25], [0, 0, 0, 1.0]])
    dst_shape = (2 * dhw[0] + 3, 2 * dhw[1] + 1, 2 * dhw[2] + 2)
    for mode, pad, tol in (("trilinear", "border", 1e-4), ("trilinear", "zeros", 1e-4), ("nearest", "border", 0.0)):
        src = img if mode == "trilinear" else lab
        got = prepost.resample_to_match(src.to(dev), a_src, a_dst, dst_shape, mode, pad, _backend=be)
        want = P.resample_to_match_ref(src, a_src, a_dst, dst_shape, mode, pad)
        if tol:
            assert float((got.cpu() - want).abs().max()) < tol, (mode, pad, float((got.cpu() - want).abs().max()))
        else:
            assert float((got.cpu() != want).float().mean()) < 1e-3, mode  # ties at exact .5 coordinates may round either way
    # identity map reproduces the input exactly
    eye = torch.eye(4)
    assert torch.equal(prepost.resample_to_match(img.to(dev), eye, eye, dhw, _backend=be).cpu(), img), "identity"


def test_prepost_on_emulator(emu_backend):
    _cases("cpu", emu_backend, (6, 7, 9))
    _resample_cases("cpu", emu_backend, (6, 7, 9))


def test_prepost_refuses_cpu():
    with pytest.raises(RuntimeError, match="MI355X"):
        prepost.normalize_intensity(torch.zeros(1, 4, 4, 4))


@pytest.mark.gpu
def test_prepost_gpu(hip_backend):
    _cases("cuda", None, (40, 33, 50))
    _resample_cases("cuda", None, (40, 33, 50))
    # the reference's own known answer at its full 100^3 size
    data = _label_volume()
    got = prepost.compile_one_hot_encoding(data.cuda(), 2, labels=[5, 22])
    assert torch.equal(got.cpu(), torch.cat([(data[0] == 5), (data[0] == 22)]).to(torch.uint8))
