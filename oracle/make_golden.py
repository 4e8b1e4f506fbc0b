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


def make_autocast(path, bundle_path):
    """The reference's UNet3D graph (same weights and input as `bundle_path`) under torch.autocast -- what its AutocastUNet
    (models/pytorch/segmentation/unet.py:53-58) adds around UNet3D.forward. The class itself enters torch.cuda.amp.autocast, which
    is a no-op without a CUDA device, so the context is entered here for the CPU device type with the two 16-bit dtypes: convolutions run
    on fp16 / bf16 operands and return that dtype, the norms run in fp32 (torch's autocast policy, the same op lists as on CUDA)."""
    b = torch.load(bundle_path)
    model = reference_shim.build_reference_unet3d(seed=1234, **b["kwargs"]).eval()
    model.load_state_dict(b["state_dict"])
    out = {"kwargs": b["kwargs"], "bundle": os.path.basename(bundle_path),
           "generator": "oracle/make_golden.py: reference UNet3D under torch.autocast('cpu', dtype), torch " + torch.__version__}
    with torch.no_grad():
        ref = model(b["x"])
        for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
            with torch.autocast("cpu", dtype=dt):
                o = model(b["x"])
            out["logits_" + name] = o.float()
            print(name, "max rel. difference to the fp32 logits:", float((o.float() - ref).abs().max() / ref.abs().max()))
    torch.save(out, path)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    make(os.path.join(out, "unet3d_small.pt"), dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 2, 1]), (20, 16, 24), 2)
    make(os.path.join(out, "unet3d_small_transposed.pt"),
         dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1, 1], use_transposed_convolutions=True), (16, 16, 16), 1)
    make_autocast(os.path.join(out, "unet3d_small_autocast.pt"), os.path.join(out, "unet3d_small.pt"))
