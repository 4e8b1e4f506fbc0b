"""Developer tool (GPU): fused-statistics run vs standalone run of a small UNet3D: logits and every gradient, against fp64 truth."""
import sys, importlib, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import torch_ops as O, unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet"); losses = importlib.import_module("3dunetcnn_amd.losses")
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
cin = 3; dhw = (16, 20, 24)
x, y = R.synthetic_case(1, cin, dhw, 2)
torch.manual_seed(3)
m0 = unet.HipUNet3D(n_features=cin, n_outputs=2, base_width=8, encoder_blocks=[1, 1]).eval()
sd = {k: v.detach().clone().double().requires_grad_(True) for k, v in m0.state_dict().items()}
ref = R.unet3d_forward(sd, x.double(), (1, 1)); l = O.dice_loss(ref, y); l.backward()
g64 = {k: v.grad for k, v in sd.items()}
out = {}
for fused in (False, True):
    be.fused_stats = fused
    torch.manual_seed(3)
    m = unet.HipUNet3D(n_features=cin, n_outputs=2, base_width=8, encoder_blocks=[1, 1]).cuda().eval()
    o = m(x.cuda()); loss = losses.HipDiceLoss(sigmoid=True)(o, y.cuda()); loss.backward(); torch.cuda.synchronize()
    out[fused] = (o.detach().cpu().double(), float(loss), {k: p.grad.cpu().double() for k, p in m.named_parameters()})
print("logits err vs fp64: plain %.2e fused %.2e ; |logits| %.2f" % (float((out[False][0] - ref.detach()).abs().max()), float((out[True][0] - ref.detach()).abs().max()), float(ref.abs().max())))
print("loss plain %.9f fused %.9f truth %.9f" % (out[False][1], out[True][1], float(l)))
gmax = max(float(v.abs().max()) for v in g64.values())
for k in g64:
    a, b = out[False][2][k], out[True][2][k]
    print("%-50s scale %.1e  plain %.1e  fused %.1e   (rel gmax: %.1e / %.1e)" % (k, float(g64[k].abs().max()), float((a - g64[k]).abs().max()), float((b - g64[k]).abs().max()),
          float((a - g64[k]).abs().max()) / gmax, float((b - g64[k]).abs().max()) / gmax))
