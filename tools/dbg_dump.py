"""Developer tool: dump statistics + gradients of the small 3-channel UNet3D (fused statistics) to a file. python tools/dbg_dump.py out.pt [emu]"""
import sys, importlib, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet"); ops = importlib.import_module("3dunetcnn_amd.ops"); losses = importlib.import_module("3dunetcnn_amd.losses")
lib_mod = importlib.import_module("3dunetcnn_amd._lib")
emu = len(sys.argv) > 2 and sys.argv[2] == "emu"
be = ops.Backend(lib=lib_mod.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "emu", "libmi355unet3d_emu.so"))), device="cpu") if emu else ops.default_backend()
dev = "cpu" if emu else "cuda"
cin = 3; dhw = (16, 20, 24)
x, y = R.synthetic_case(1, cin, dhw, 2)
res = {}
for fused in (False, True):
    be.fused_stats = fused
    stats = []
    og = be.gn_stats
    def pg(*a, **k):
        r = og(*a, **k); stats.append([t.detach().cpu().clone() for t in r]); return r
    be.gn_stats = pg
    ob = be.gn_act_bwd
    bw = []
    def pb(xa, dA, dx, groups, slope, gamma, mr, sc, sh, dg, db, addend=None, partials=None):
        dA_in = dA.tensor().detach().cpu().clone()
        r = ob(xa, dA, dx, groups, slope, gamma, mr, sc, sh, dg, db, addend=addend, partials=partials)
        bw.append(dict(dA=dA_in, dx=dx.tensor().detach().cpu().clone(), dg=dg.detach().cpu().clone(), db=db.detach().cpu().clone(), x=xa.tensor().detach().cpu().clone(),
                       sc=sc.detach().cpu().clone(), sh=sh.detach().cpu().clone(), mr=mr.detach().cpu().clone()))
        return r
    be.gn_act_bwd = pb
    torch.manual_seed(3)
    m = unet.HipUNet3D(n_features=cin, n_outputs=2, base_width=8, encoder_blocks=[1, 1]).to(dev).eval()
    m._be = be
    crit = losses.HipDiceLoss(sigmoid=True); crit._be = be
    o = m(x.to(dev)); loss = crit(o, y.to(dev)); loss.backward()
    be.gn_stats = og; be.gn_act_bwd = ob
    res[fused] = dict(stats=stats, bw=bw, logits=o.detach().cpu(), grads={k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
torch.save(res, sys.argv[1])
