"""Developer tool: checks each gn_act_bwd / dgrad call inside a real backward against fp64 torch math (needs an MI355X)."""
import importlib, os, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import op_cases as C
from oracle import unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet")
losses = importlib.import_module("3dunetcnn_amd.losses")
ops = importlib.import_module("3dunetcnn_amd.ops")

be = ops.default_backend()
orig_gn = be.gn_act_bwd
orig_conv = be.conv_fwd
count = {"gn": 0, "conv": 0}

def gn_wrap(x, dA, dx, groups, slope, gamma, mean_rstd, scale, shift, dgamma, dbeta, addend=None):
    xt = x.tensor().double(); dat = dA.tensor().double().clone()
    addt = addend.tensor().double().clone() if addend is not None else None
    orig_gn(x, dA, dx, groups, slope, gamma, mean_rstd, scale, shift, dgamma, dbeta, addend=addend)
    if count["gn"] < 8:
        n, d, h, w, c = xt.shape
        u = xt * scale.double()[:, None, None, None, :] + shift.double()[:, None, None, None, :]
        du = dat * (u > 0)
        cpg = c // groups
        mean = mean_rstd[..., 0].double().repeat_interleave(cpg, 1)[:, None, None, None, :]
        rstd = mean_rstd[..., 1].double().repeat_interleave(cpg, 1)[:, None, None, None, :]
        xh = (xt - mean) * rstd
        s1 = du.sum((0, 1, 2, 3)); s2 = (du * xh).sum((0, 1, 2, 3))
        # exact stats check
        xg = xt.reshape(n, -1, groups, cpg)
        m_ref = xg.mean((1, 3)); v_ref = xg.var((1, 3), unbiased=False)
        g = gamma.double()[None, None, None, None, :]
        gd = (g * du).reshape(n, -1, groups, cpg); gdx = (g * du * xh).reshape(n, -1, groups, cpg)
        m1 = gd.mean((1, 3)).repeat_interleave(cpg, 1)[:, None, None, None, :]; m2 = gdx.mean((1, 3)).repeat_interleave(cpg, 1)[:, None, None, None, :]
        dxr = rstd * (g * du - m1 - xh * m2)
        if addt is not None: dxr = dxr + addt
        print(f"gn#{count['gn']} C={c} G={groups} V={d*h*w}: dbeta {C.rel_err(dbeta, s1):.2e} dgamma {C.rel_err(dgamma, s2):.2e} dx {C.rel_err(dx.tensor(), dxr):.2e} "
              f"mean {C.rel_err(mean_rstd[...,0], m_ref):.2e} rstd {C.rel_err(mean_rstd[...,1], (v_ref+1e-5).rsqrt()):.2e} |dA|max {float(dat.abs().max()):.2e} sum|du| {float(du.abs().sum((0,1,2,3)).max()):.2e} max|s1| {float(s1.abs().max()):.2e}", flush=True)
    count["gn"] += 1

be.gn_act_bwd = gn_wrap
torch.manual_seed(1234)
m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().eval()
x, y = R.synthetic_case(1, 4, (64, 64, 64), 3)
crit = losses.HipDiceLoss(sigmoid=True)
out = m(x.cuda()); loss = crit(out, y.cuda()); loss.backward()
