"""Developer tool (CPU): numerical probe for a 3-D Winograd F(2x2x2, 3x3x3) convolution in fp32 (tools/NEXT.md). Computes
conv3d(x, w, padding=1) through the transform domain with fp32 arithmetic and compares it -- and the direct fp32 convolution -- with
the fp64 direct result; and the weight gradient through F(3x3, 2x2) in the plane x direct z (the form built on branch winograd-prep).
    python tools/winograd_probe.py [cin] [cout] [size]"""
import sys

import torch
import torch.nn.functional as F

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 16
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)      # input transform (4x4)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)          # filter transform (4x3)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)                                     # output transform (2x4)


def t3(m, t, dims):
    """apply the 1-D transform m along each of the three dims of t"""
    for d in dims:
        t = torch.movedim(torch.tensordot(m.to(t.dtype), torch.movedim(t, d, 0), dims=([1], [0])), 0, d)
    return t


def winograd(x, w, dtype):
    n, ci, D, H, W = x.shape
    xp = F.pad(x.to(dtype), (1, 1, 1, 1, 1, 1))
    # tiles: 4x4x4 windows with stride 2 -> [n, ci, tz, ty, tx, 4, 4, 4]
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2).unfold(4, 4, 2)
    V = t3(BT, t, (5, 6, 7))                                     # transformed input
    U = t3(G, w.to(dtype), (2, 3, 4))                            # transformed filter [co, ci, 4, 4, 4]
    M = torch.einsum("nczyxabd,ocabd->nozyxabd", V, U)          # 64 point-wise channel GEMMs (the MFMA part)
    Y = t3(AT, M, (5, 6, 7))                                     # [n, co, tz, ty, tx, 2, 2, 2]
    tz, ty, tx = Y.shape[2:5]
    return Y.permute(0, 1, 2, 5, 3, 6, 4, 7).reshape(n, w.shape[0], 2 * tz, 2 * ty, 2 * tx)


torch.manual_seed(0)
x = torch.randn(1, cin, S, S, S)
w = torch.randn(cout, cin, 3, 3, 3) / (27 * cin) ** 0.5
ref = F.conv3d(x.double(), w.double(), padding=1)
d32 = F.conv3d(x, w, padding=1)
w32 = winograd(x, w, torch.float32)
w64 = winograd(x, w, torch.float64)
scale = ref.abs().max()
print(f"cin {cin} cout {cout} size {S}^3; max |error| / max |y| against the fp64 direct convolution:")
print(f"  Winograd in fp64          {float((w64 - ref).abs().max() / scale):.2e}   (algebra check)")
print(f"  direct conv in fp32       {float((d32.double() - ref).abs().max() / scale):.2e}")
print(f"  Winograd F(2,3)^3 in fp32 {float((w32.double() - ref).abs().max() / scale):.2e}")
print(f"  rms ratio Winograd / direct: {float((w32.double() - ref).pow(2).mean().sqrt() / (d32.double() - ref).pow(2).mean().sqrt()):.2f}")


# ---- weight gradient: F(3x3, 2x2) in the (y, x) plane, direct along z (what conv3d_wino2d_wgrad computes) ----
A2 = torch.tensor([[1, 0], [1, 1], [1, -1], [0, -1]], dtype=torch.float64)                  # dy-tile transform (4x2)
GT = torch.tensor([[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]], dtype=torch.float64)    # tap transform (3x4)


def winograd_wgrad(x, dy, dtype):
    n, ci, D, H, W = x.shape
    xp = F.pad(x.to(dtype), (1, 1, 1, 1, 1, 1))
    dw = torch.zeros(dy.shape[1], ci, 3, 3, 3, dtype=dtype)
    h = t3(A2, dy.to(dtype).unfold(3, 2, 2).unfold(4, 2, 2), (5, 6))
    for dz in range(3):
        V = t3(BT, xp[:, :, dz:dz + D].unfold(3, 4, 2).unfold(4, 4, 2), (5, 6))
        dw[:, :, dz] = t3(GT, torch.einsum("nozyxab,nczyxab->ocab", h, V), (2, 3))
    return dw


xa = torch.randn(2, cin, S, S, S).relu() + 0.1 * torch.randn(2, cin, S, S, S)       # post-activation-like input
dy = torch.randn(2, cout, S, S, S) * 1e-3
w0 = torch.zeros(cout, cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
(gref,) = torch.autograd.grad(F.conv3d(xa.double(), w0, padding=1), w0, dy.double())
w1 = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
(g32,) = torch.autograd.grad(F.conv3d(xa, w1, padding=1), w1, dy)
sc = gref.abs().max()
print("weight gradient over 2 x size^3 voxels, max |error| / max |dw| against the fp64 direct gradient:")
print(f"  Winograd in fp64          {float((winograd_wgrad(xa, dy, torch.float64) - gref).abs().max() / sc):.2e}   (algebra check)")
print(f"  direct wgrad in fp32      {float((g32.double() - gref).abs().max() / sc):.2e}")
print(f"  Winograd F(3,2)^2 in fp32 {float((winograd_wgrad(xa, dy, torch.float32).double() - gref).abs().max() / sc):.2e}")

