"""TEST INFRASTRUCTURE (oracle) -- generates tests/golden/unet3d_small.pt by importing and running the REFERENCE's own
UNet3D from /root/reference (build container only; see oracle/reference_shim.py). Commit the output.

    python oracle/make_golden.py

Contents: constructor kwargs, the reference's state_dict under seed 1234, a synthetic input/target,
the reference's logits (eval mode: Dropout3d off, myronenko.py:70-79), sigmoid-Dice loss (oracle/torch_ops.py) and every
parameter gradient -- fp32 CPU.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import reference_shim, torch_ops, unet3d_ref  # noqa: E402


def make(path, kwargs, dhw, n):
    model = reference_shim.build_reference_unet3d(seed=1234, **kwargs).eval()
    x, y = unet3d_ref.synthetic_case(n, kwargs["n_features"], dhw, kwargs["n_outputs"], seed=0)
    logits = model(x)
    loss = torch_ops.dice_loss(logits, y)
    loss.backward()
    bundle = {
        "kwargs": kwargs, "state_dict": {k: v.detach().clone() for k, v in model.state_dict().items()},
        "x": x, "y": y, "logits": logits.detach(), "loss": loss.detach(),
        "grads": {k: p.grad.detach().clone() for k, p in model.named_parameters()},
        "generator": "oracle/make_golden.py: reference unet3d.models.pytorch.segmentation.unet.UNet3D, torch "
                     + torch.__version__ + " CPU fp32",
    }
    torch.save(bundle, path)
    print(path, os.path.getsize(path) // 1024, "KiB; loss", float(loss))


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    make(os.path.join(out, "unet3d_small.pt"), dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 2, 1]), (20, 16, 24), 2)
    make(os.path.join(out, "unet3d_small_transposed.pt"),
         dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1, 1], use_transposed_convolutions=True), (16, 16, 16), 1)
