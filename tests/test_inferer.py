"""HipSlidingWindowInferer (drop-in for MONAI SlidingWindowInferer, BASELINE configs[4]) vs the plain-torch restatement
oracle/sliding_window_ref.py. CPU legs run the HIP accumulate/normalise kernels on the emulator with a torch predictor; GPU legs
run them on the device, with a torch predictor (exactness of the driver) and with HipUNet3D as the network."""
import importlib

import pytest
import torch
import torch.nn.functional as F

import op_cases as C
from oracle import sliding_window_ref as S, unet3d_ref as R

inferer = importlib.import_module("3dunetcnn_amd.inferer")
unet = importlib.import_module("3dunetcnn_amd.unet")


def test_window_plan_known_answer():
    # 240x240x155, 128^3 windows, overlap 0.5 -> starts {0,64,112}^2 x {0,27} = 18 windows
    st = inferer._window_starts([240, 240, 155], [128, 128, 128], inferer._scan_interval([240, 240, 155], [128, 128, 128], (0.5,) * 3))
    assert len(st) == 18
    assert sorted({s[0] for s in st}) == [0, 64, 112] and sorted({s[2] for s in st}) == [0, 27]
    st = inferer._window_starts([240, 240, 155], [128, 128, 128], inferer._scan_interval([240, 240, 155], [128, 128, 128], (0.25,) * 3))
    assert len(st) == 18 and sorted({s[0] for s in st}) == [0, 96, 112]
    assert st == S.dense_patch_starts([240, 240, 155], [128] * 3, S.scan_interval([240, 240, 155], [128] * 3, 0.25))


def _predictor(seed=0):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(3, 4, 3, 3, 3, generator=g) * 0.2

    def f(x):
        return F.conv3d(x, w.to(x.device), padding=1)
    return f


@pytest.mark.parametrize("kw", [
    dict(size=(20, 17, 13), roi=(8, 8, 8), overlap=0.5, mode="constant", bs=4),
    dict(size=(11, 16, 9), roi=(8, 8, 8), overlap=0.25, mode="gaussian", bs=3),
    dict(size=(6, 9, 7), roi=(8, 8, 8), overlap=0.25, mode="gaussian", bs=2),       # volume smaller than the window: padded
])
def test_inferer_matches_oracle_on_emulator(emu_backend, kw):
    x = torch.randn(2, 4, *kw["size"], generator=torch.Generator().manual_seed(1))
    inf = inferer.HipSlidingWindowInferer(kw["roi"], sw_batch_size=kw["bs"], overlap=kw["overlap"], mode=kw["mode"])
    inf._be = emu_backend
    got = inf(x, _predictor())
    ref = S.sliding_window_inference(x, kw["roi"], kw["bs"], _predictor(), kw["overlap"], kw["mode"])
    assert got.shape == ref.shape
    assert C.rel_err(got, ref) < 1e-6


def test_inferer_rejects_unknown_options():
    with pytest.raises(NotImplementedError):
        inferer.HipSlidingWindowInferer((8, 8, 8), padding_mode="reflect")
    with pytest.raises(RuntimeError, match="MI355X"):
        inferer.HipSlidingWindowInferer((8, 8, 8))(torch.zeros(1, 4, 8, 8, 8), _predictor())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["constant", "gaussian"])
def test_inferer_gpu_torch_predictor(mode):
    x = torch.randn(1, 4, 70, 61, 50, generator=torch.Generator().manual_seed(2))
    inf = inferer.HipSlidingWindowInferer((32, 32, 32), sw_batch_size=4, overlap=0.5, mode=mode)
    got = inf(x.cuda(), _predictor())
    ref = S.sliding_window_inference(x, (32, 32, 32), 4, _predictor(), 0.5, mode)
    assert C.rel_err(got, ref) < 1e-5


@pytest.mark.gpu
def test_inferer_gpu_with_hip_unet():
    torch.manual_seed(5)
    m = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=16, encoder_blocks=[1, 1, 2]).cuda().eval()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x = torch.randn(1, 4, 60, 72, 50, generator=torch.Generator().manual_seed(3))
    inf = inferer.HipSlidingWindowInferer((32, 32, 32), sw_batch_size=2, overlap=0.25, mode="gaussian")
    got = inf(x.cuda(), m)
    with torch.no_grad():
        ref = S.sliding_window_inference(x, (32, 32, 32), 2, lambda w: R.unet3d_forward(sd, w, (1, 1, 2)), 0.25, "gaussian")
    assert C.rel_err(got, ref) < 1e-3
