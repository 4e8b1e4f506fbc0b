"""Developer tool (GPU): ReLU ties of the last decoder block between the fused-statistics and the standalone-statistics forward."""
import sys, importlib, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet"); ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
cin = 3; dhw = (16, 20, 24)
x, y = R.synthetic_case(1, cin, dhw, 2)
sv = {}
for fused in (False, True):
    be.fused_stats = fused
    torch.manual_seed(3)
    m = unet.HipUNet3D(n_features=cin, n_outputs=2, base_width=8, encoder_blocks=[1, 1]).cuda().eval()
    m.flatten_parameters()
    with torch.no_grad():
        logits, saved = m._forward_impl(x.cuda().contiguous(), True)
    torch.cuda.synchronize()
    sv[fused] = saved
for name, get in (("last.h1/st2", lambda s: (s["last"][0].h1, s["last"][0].st2)), ("last.x/st1", lambda s: (s["last"][0].x, s["last"][0].st1)),
                  ("enc0.h1/st2", lambda s: (s["enc"][0][0].h1, s["enc"][0][0].st2))):
    (hp, stp), (hf, stf) = get(sv[False]), get(sv[True])
    tp, tf = hp.tensor(), hf.tensor()
    print(name, "tensor bitwise equal:", bool(torch.equal(tp, tf)), "| scale max rel diff %.2e  shift max abs diff %.2e" % (
        float(((stp[1] - stf[1]) / stp[1]).abs().max()), float((stp[2] - stf[2]).abs().max())))
    up = tp * stp[1][:, None, None, None, :] + stp[2][:, None, None, None, :]
    uf = tf * stf[1][:, None, None, None, :] + stf[2][:, None, None, None, :]
    print("   mask flips:", int(((up > 0) != (uf > 0)).sum()), " |u|<1e-6:", int((up.abs() < 1e-6).sum()), " |u|<1e-5:", int((up.abs() < 1e-5).sum()), " exact zeros:", int((up == 0).sum()),
          " numel", up.numel(), " max|u_p - u_f| %.2e" % float((up - uf).abs().max()))
    print("   mean_rstd plain", stp[0].flatten()[:6].tolist()); print("   mean_rstd fused", stf[0].flatten()[:6].tolist())
