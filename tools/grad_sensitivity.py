"""Developer tool (CPU only): shows that the UNet3D parameter gradients are ill-conditioned w.r.t. fp32 rounding.

Runs the CPU oracle graph in fp64 (truth) and in fp32 with relative Gaussian noise injected into every conv3d output.
Noise of 1e-6 (the size of a k-ordered fp32 accumulation over K~3.5k terms) already moves some encoder gradients by
4e-3 -- the same tensors, by the same amount, as the MI355X path (profiles/ and DESIGN.md, 'Gradient conditioning').
"""
import sys, importlib
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
import op_cases as C
from oracle import torch_ops as O, unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet")
torch.manual_seed(1234)
m = unet.HipUNet3D(n_features=4, n_outputs=3)
dhw=(64,64,64)
x, y = R.synthetic_case(1, 4, dhw, 3)
def run(dt, noise=0.0):
    sd = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
    orig = F.conv3d
    g = torch.Generator().manual_seed(0)
    def noisy(*a, **k):
        o = orig(*a, **k)
        if noise: o = o * (1 + noise*torch.randn(o.shape, generator=g, dtype=o.dtype))
        return o
    R.F.conv3d = noisy
    try:
        out = R.unet3d_forward(sd, x.to(dt)); l = O.dice_loss(out, y); l.backward()
    finally:
        R.F.conv3d = orig
    return out.detach(), {k: v.grad for k,v in sd.items()}
o64,g64 = run(torch.float64)
for noise in (0.0, 3e-7, 1e-6):
    o,g = run(torch.float32, noise)
    errs = sorted(((C.rel_err(g[k], g64[k]),k) for k in g), reverse=True)
    print('noise',noise,'logits',C.rel_err(o,o64),'worst grads', [(f"{e:.1e}",k.replace('encoder.layers','enc').replace('decoder.layers','dec')) for e,k in errs[:4]], 'median', f"{errs[len(errs)//2][0]:.1e}")
