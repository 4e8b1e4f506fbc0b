"""TEST INFRASTRUCTURE (oracle) -- CPU fp32 restatement of MONAI `DynUNet` as configured by the reference's shipped configs
(examples/brats2020/brats2020_config.json:2-107; model lookup unet3d/models/build.py:9-13 through
unet3d/models/pytorch/__init__.py:1 `from monai.networks.nets import *`), as a pure function of a state_dict.

PARITY UNPINNED: MONAI is an un-vendored, unpinned third-party dependency of the reference (requirements.txt:4) and is not
installed in this environment (no network), so this restatement cannot be checked against MONAI itself; the reference's
own tests hold no fixture for the model either. It restates the published MONAI >= 1.2 structure
(monai/networks/nets/dynunet.py, monai/networks/blocks/dynunet_block.py):

  UnetBasicBlock : Conv3d(k3, stride s, pad 1, bias=False) -> InstanceNorm3d(affine, eps 1e-5) -> LeakyReLU(0.01), twice
  UnetUpBlock    : ConvTranspose3d(k2, s2, bias=False) -> cat((up, skip), 1) -> UnetBasicBlock(2c, c, stride 1)
  UnetOutBlock   : Conv3d(k1, bias=True)
  forward        : input_block -> downsamples -> bottleneck -> upsamples (deepest first) -> output_block
"""
import torch
import torch.nn.functional as F


def _basic_block(sd, pre, x, stride):
    x = F.conv3d(x, sd[pre + ".conv1.conv.weight"], None, stride=stride, padding=1)
    x = F.leaky_relu(F.instance_norm(x, None, None, sd[pre + ".norm1.weight"], sd[pre + ".norm1.bias"], True, 0.1, 1e-5), 0.01)
    x = F.conv3d(x, sd[pre + ".conv2.conv.weight"], None, stride=1, padding=1)
    x = F.leaky_relu(F.instance_norm(x, None, None, sd[pre + ".norm2.weight"], sd[pre + ".norm2.bias"], True, 0.1, 1e-5), 0.01)
    return x


def dynunet_forward(sd, x, n_levels):
    """sd: state_dict with MONAI's key names; x: [N, C, D, H, W]; n_levels = len(filters)."""
    skips = [_basic_block(sd, "input_block", x, 1)]
    for i in range(n_levels - 2):
        skips.append(_basic_block(sd, f"downsamples.{i}", skips[-1], 2))
    x = _basic_block(sd, "bottleneck", skips[-1], 2)
    for k in range(n_levels - 1):
        x = F.conv_transpose3d(x, sd[f"upsamples.{k}.transp_conv.conv.weight"], None, stride=2)
        x = torch.cat((x, skips[n_levels - 2 - k]), 1)
        x = _basic_block(sd, f"upsamples.{k}.conv_block", x, 1)
    return F.conv3d(x, sd["output_block.conv.conv.weight"], sd["output_block.conv.conv.bias"])
