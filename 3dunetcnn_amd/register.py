"""register(): make the MI355X modules discoverable through the reference's own lookup paths, unchanged:

  * models:     getattr(unet3d.models.pytorch, model_name)(**kwargs)   (unet3d/models/build.py:9-13)
  * criteria:   getattr(unet3d.losses, name) first                     (unet3d/scripts/script_utils.py:61-77)
  * optimizers: getattr(torch.optim, name)                             (unet3d/scripts/script_utils.py:80-81)
  * inferers:   getattr(monai.inferers, name)                          (unet3d/scripts/script_utils.py:290-293)

so a config with {"model": {"name": "HipUNet3D", ...}, "loss": {"name": "HipDiceLoss", ...},
"optimizer": {"name": "HipAdam", ...}} trains through unet3d.train.run_training as is. With replace=True the
reference names themselves ("UNet3D", "DynUNet", "DiceLoss") resolve to the MI355X implementations.
"""
import importlib

import torch


def register(replace=False):
    from .dynunet import HipDynUNet
    from .losses import HipBCEWithLogitsLoss, HipCrossEntropyLoss, HipDiceCELoss, HipDiceLoss, HipGeneralizedDiceLoss
    from .optim import HipAdam
    from .unet import HipAutocastUNet, HipAutoImplantUNet, HipUNet3D
    done = {}
    try:
        models = importlib.import_module("unet3d.models.pytorch")
        models.HipUNet3D = HipUNet3D
        models.HipDynUNet = HipDynUNet
        models.HipAutocastUNet = HipAutocastUNet
        models.HipAutoImplantUNet = HipAutoImplantUNet
        done["models"] = ["HipUNet3D", "HipDynUNet", "HipAutocastUNet", "HipAutoImplantUNet"]
        if replace:
            models.UNet3D = HipUNet3D
            models.DynUNet = HipDynUNet
            models.AutocastUNet = HipAutocastUNet
            models.AutoImplantUNet = HipAutoImplantUNet
            done["models"] += ["UNet3D", "DynUNet", "AutocastUNet", "AutoImplantUNet"]
    except ImportError:
        done["models"] = []
    try:
        losses = importlib.import_module("unet3d.losses")
        hip_losses = {"HipDiceLoss": HipDiceLoss, "HipDiceCELoss": HipDiceCELoss, "HipGeneralizedDiceLoss": HipGeneralizedDiceLoss,
                      "HipBCEWithLogitsLoss": HipBCEWithLogitsLoss,
                      "HipCrossEntropyLoss": HipCrossEntropyLoss}
        for name, cls in hip_losses.items():
            setattr(losses, name, cls)
        done["losses"] = list(hip_losses)
        if replace:
            # unet3d.losses is searched before torch.nn and monai.losses (script_utils.py:61-77), so these shadow them
            for name, cls in hip_losses.items():
                setattr(losses, name[3:], cls)
                done["losses"].append(name[3:])
    except ImportError:
        done["losses"] = []
    torch.optim.HipAdam = HipAdam
    done["optim"] = ["HipAdam"]
    try:
        from .inferer import HipSlidingWindowInferer
        inferers = importlib.import_module("monai.inferers")
        inferers.HipSlidingWindowInferer = HipSlidingWindowInferer
        done["inferers"] = ["HipSlidingWindowInferer"]
        if replace:
            inferers.SlidingWindowInferer = HipSlidingWindowInferer
            done["inferers"].append("SlidingWindowInferer")
    except ImportError:
        done["inferers"] = []
    return done
