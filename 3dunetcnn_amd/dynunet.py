"""HipDynUNet placeholder import target (implemented in dynunet_impl once the UNet3D path is measured)."""


class HipDynUNet:  # replaced below when the implementation module is importable
    def __init__(self, *a, **k):
        raise NotImplementedError("HipDynUNet is not implemented yet in this build")
