"""Developer tool (GPU): per gn_stats call of a small UNet3D forward, fused vs standalone statistics."""
import sys, importlib, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet"); ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dhw = (16, 20, 24)
x, y = R.synthetic_case(1, cin, dhw, 2)
orig = be.gn_stats
trace = {}
for fused in (True, False):
    be.fused_stats = fused
    rec = []
    def patched(xa, groups, eps, gamma, beta, _rec=rec):
        had = xa.mom is not None
        out = orig(xa, groups, eps, gamma, beta)
        # also the standalone statistics of the same tensor, right now
        saved, xa.mom = xa.mom, None
        f, be.fused_stats = be.fused_stats, False
        ref = orig(xa, groups, eps, gamma, beta)
        be.fused_stats = f; xa.mom = saved
        t = xa.tensor().detach().cpu().double()
        n_, c_ = t.shape[0], t.shape[-1]
        tg = t.permute(0, 4, 1, 2, 3).reshape(n_, groups, -1)
        truth = torch.stack((tg.mean(-1), (tg.var(-1, unbiased=False) + eps).rsqrt()), -1)
        _rec.append((tuple(xa.shape), groups, had, out[0].clone().cpu().double(), ref[0].clone().cpu().double(), None if saved is None else [(m[1], m[2]) for m in saved], truth))
        return out
    be.gn_stats = patched
    torch.manual_seed(3)
    m = unet.HipUNet3D(n_features=cin, n_outputs=2, base_width=8, encoder_blocks=[1, 1]).cuda().eval()
    with torch.no_grad():
        m(x.cuda())
    torch.cuda.synchronize()
    trace[fused] = rec
be.gn_stats = orig
for i, (a, b) in enumerate(zip(trace[True], trace[False])):
    def err(st, truth):      # (mean error in units of sigma, relative rstd error)
        return float(((st[..., 0] - truth[..., 0]) * truth[..., 1]).abs().max()), float(((st[..., 1] - truth[..., 1]) / truth[..., 1]).abs().max())
    ef, es = err(a[3], a[6]), err(a[4], a[6])
    print(i, a[0], "groups", a[1], "records", a[5], "| fused: mean %.1e sigma, rstd %.1e | standalone: mean %.1e sigma, rstd %.1e" % (ef + es), flush=True)
