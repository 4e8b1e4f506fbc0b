"""TEST INFRASTRUCTURE (oracle) -- imports the REFERENCE's own UNet3D from /root/reference, untouched.

Only usable in the build container (the GPU box has no /root/reference): it validates the restatement
(oracle/unet3d_ref.py) and generates the golden vectors committed under tests/golden/ (oracle/make_golden.py).

Shim (verified):
  * `monai` is not installed -> empty stub modules satisfy `from monai.networks.nets import *`
    (unet3d/models/pytorch/__init__.py:1);
  * unet3d/models/pytorch/segmentation/unet.py:38 uses `F.pad` without importing F -> inject torch.nn.functional.
Reference files are never modified or copied.
"""
import contextlib
import io
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MI355_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "unet3d"))


def import_reference_unet():
    if not available():
        raise RuntimeError(f"{REFERENCE_ROOT} not present (the reference only exists in the build container)")
    import torch
    for n in ("monai", "monai.networks", "monai.networks.nets"):
        if n not in sys.modules:
            mod = types.ModuleType(n)
            mod.__all__ = []
            sys.modules[n] = mod
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from unet3d.models.pytorch.segmentation import unet
    unet.F = torch.nn.functional
    return unet


def build_reference_unet3d(seed=1234, **kwargs):
    """Reference UNet3D with its default init under `seed`; constructor prints are swallowed."""
    import torch
    unet = import_reference_unet()
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        model = unet.UNet3D(**kwargs)
    return model
