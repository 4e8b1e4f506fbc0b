"""TEST INFRASTRUCTURE (oracle) -- CPU fp32 restatement of the reference UNet3D forward as a pure function of a
state_dict (torch ATen on CPU; autograd provides the gradient oracle).

Pinned against the reference itself: tests/test_oracle_pinned.py imports /root/reference's UNet3D (oracle/reference_shim.py)
and requires bit-identical logits and gradients from this restatement on several configurations, and the committed
golden vectors (tests/golden/unet3d_small.pt, made by oracle/make_golden.py from the imported reference) pin it on the
GPU box where /root/reference does not exist.

Follows:
  unet3d/models/pytorch/segmentation/unet.py:7-16   UNetEncoder.forward (skip list)
  unet3d/models/pytorch/segmentation/unet.py:27-44  UNetDecoder.forward (layer -> pre -> up -> pad -> cat)
  unet3d/models/pytorch/classification/myronenko.py:17-21, 47-58, 75-80  (GN->ReLU->conv, residual, Dropout3d)
  unet3d/models/pytorch/classification/decoder.py:99-106 (ConvTranspose3d k3 s2 p1 | 1x1x1 + trilinear x2)
  unet3d/models/pytorch/autoencoder/variational.py:81-87 (encoder -> decoder -> final 1x1x1 -> activation)
"""
import importlib

import torch
import torch.nn.functional as F


def _groups(planes, norm_groups=8):
    # myronenko.py:23-31
    if planes < norm_groups or planes % norm_groups:
        return planes
    return norm_groups


def _conv_block(sd, pre, x):
    g = sd[pre + ".norm1.weight"]
    x = F.group_norm(x, _groups(g.numel()), g, sd[pre + ".norm1.bias"], 1e-5)
    x = F.relu(x)
    return F.conv3d(x, sd[pre + ".conv.weight"], None, stride=1, padding=1)


def _res_block(sd, pre, x):
    identity = x
    y = _conv_block(sd, pre + ".conv1", x)
    y = _conv_block(sd, pre + ".conv2", y)
    if pre + ".sample.weight" in sd:
        identity = F.conv3d(identity, sd[pre + ".sample.weight"])
    return y + identity


def _layer(sd, pre, x, n_blocks, dropout_scale=None):
    for j in range(n_blocks):
        x = _res_block(sd, f"{pre}.blocks.{j}", x)
        if j == 0 and dropout_scale is not None:
            x = x * dropout_scale[:, :, None, None, None]      # Dropout3d in train mode with an explicit mask/(1-p)
    return x


def unet3d_forward(sd, x, encoder_blocks=(1, 2, 2, 4), decoder_blocks=None, use_transposed_convolutions=False,
                   activation=None, dropout_scale=None):
    """sd: state_dict with the reference's key names; x: [N, C, D, H, W] fp32 on CPU.
    dropout_scale: None (eval mode) or [N, base_width] tensor = keep_mask/(1-p) for encoder level 0 block 0."""
    L = len(encoder_blocks)
    decoder_blocks = decoder_blocks or [1] * L
    skips = []
    for i in range(L):
        x = _layer(sd, f"encoder.layers.{i}", x, encoder_blocks[i], dropout_scale if i == 0 else None)
        skips.insert(0, x)
        if i != L - 1:
            x = F.conv3d(x, sd[f"encoder.downsampling_convolutions.{i}.weight"], None, stride=2, padding=1)
    x = skips[0]
    for k in range(L - 1):
        x = _layer(sd, f"decoder.layers.{k}", x, decoder_blocks[k])
        if use_transposed_convolutions:
            x = F.conv_transpose3d(x, sd[f"decoder.upsampling_blocks.{k}.weight"], sd[f"decoder.upsampling_blocks.{k}.bias"],
                                   stride=2, padding=1)
        else:
            x = F.conv3d(x, sd[f"decoder.pre_upsampling_blocks.{k}.weight"])
            x = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)
        s = skips[k + 1]
        dz, dy, dx = s.shape[2] - x.shape[2], s.shape[3] - x.shape[3], s.shape[4] - x.shape[4]
        x = F.pad(x, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2, dz // 2, dz - dz // 2])
        x = torch.cat((x, s), 1)
    x = _layer(sd, f"decoder.layers.{L - 1}", x, decoder_blocks[L - 1])
    x = F.conv3d(x, sd["final_convolution.weight"])
    if activation == "sigmoid":
        x = torch.sigmoid(x)
    elif activation == "softmax":
        x = torch.softmax(x, dim=1)
    return x


# the synthetic batch generator of SURVEY.md 8(d) lives in the package (bench.py's measured path must not import oracle/);
# re-exported here for the tests
synthetic_case = importlib.import_module("3dunetcnn_amd.synthetic").synthetic_case
