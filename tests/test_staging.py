"""Input-side staging: DeviceStager yields the loader's batches device-resident, in order, with the on-device
z-score / one-hot prologue equal to the oracle's (oracle/prepost_ref.py)."""
import importlib

import pytest
import torch

from oracle import prepost_ref as P

staging = importlib.import_module("3dunetcnn_amd.staging")


def _batches(n, g):
    out = []
    for i in range(n):
        img = torch.randn(2, 4, 12, 10, 14, generator=g) * (i + 1) + 3 * i
        lab = torch.randint(0, 5, (2, 1, 12, 10, 14), generator=g).float()
        out.append({"image": img, "label": lab, "idx": i})
    return out


def test_stager_refuses_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="MI355X"):
        staging.DeviceStager([])


@pytest.mark.gpu
def test_stager_passes_batches_through_in_order():
    data = _batches(5, torch.Generator().manual_seed(0))
    st = staging.DeviceStager(data)
    assert len(st) == 5
    seen = 0
    for i, b in enumerate(st):
        assert b["idx"] == i and b["image"].is_cuda and b["label"].is_cuda
        # consume on the compute stream while the next upload is in flight
        s = (b["image"] * 2).sum()
        assert torch.allclose(s.cpu(), (data[i]["image"] * 2).sum(), rtol=1e-5)
        assert torch.equal(b["label"].cpu(), data[i]["label"])
        seen += 1
    assert seen == 5


@pytest.mark.gpu
def test_stager_device_prologue_matches_oracle():
    data = _batches(3, torch.Generator().manual_seed(1))
    groups = [[1, 2, 4], [1, 4], [4]]
    for i, b in enumerate(staging.DeviceStager(data, normalize=True, one_hot_labels=groups)):
        for n in range(2):
            assert torch.allclose(b["image"][n].cpu(), P.normalize_intensity(data[i]["image"][n]), atol=2e-5)
            want = P.compile_one_hot_encoding(data[i]["label"][n:n + 1], 3, labels=groups)
            assert b["label"].dtype == torch.uint8 and torch.equal(b["label"][n].cpu(), want)
