"""Developer tool: error of fused vs standalone norm statistics against double truth, per conv configuration. python tools/dbg_moments.py [emu]"""
import sys, importlib, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import op_cases as C
ops = importlib.import_module("3dunetcnn_amd.ops"); lib_mod = importlib.import_module("3dunetcnn_amd._lib")
if len(sys.argv) > 1 and sys.argv[1] == "emu":
    be = ops.Backend(lib=lib_mod.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "emu", "libmi355unet3d_emu.so"))), device="cpu")
else:
    be = ops.default_backend()
import torch.nn.functional as F
def run(n, cin, cout, dhw, stride=1, groups=None):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, *dhw, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (cin * 27) ** 0.5)
    od = [(s + 2 - 3) // stride + 1 for s in dhw]
    xa = C.to_act(be, x)
    res = {}
    for fused in (True, False):
        be.fused_stats = fused
        ya = C.to_act(be, torch.zeros(n, cout, *od))
        be.conv_fwd(xa, be.pack_weight(C.dev(be, wt), 0), ya, 3, stride, 1, moments=True)
        y = C.from_act(ya).double()
        go = groups or (8 if cout % 8 == 0 else cout)
        mr, sc, sh = be.gn_stats(ya, go, 1e-5, None, None)
        rg = y.reshape(n, go, -1)
        mean_ref, rstd_ref = rg.mean(-1), (rg.var(-1, unbiased=False) + 1e-5).rsqrt()
        res[fused] = (float((mr[..., 0].cpu().double() - mean_ref).abs().max()), float(((mr[..., 1].cpu().double() - rstd_ref) / rstd_ref).abs().max()), ya.mom is not None)
    be.fused_stats = True
    print(f"n{n} {cin}->{cout} {dhw} s{stride} g{groups}: fused(mean abs {res[True][0]:.2e}, rstd rel {res[True][1]:.2e}, rec {res[True][2]})  standalone(mean {res[False][0]:.2e}, rstd {res[False][1]:.2e})", flush=True)
run(1, 4, 8, (16, 20, 24))
run(1, 8, 8, (16, 20, 24))
run(1, 8, 8, (16, 20, 24), groups=8)
run(1, 8, 8, (16, 20, 24), stride=2)
run(1, 8, 16, (8, 10, 12))
run(1, 32, 32, (16, 16, 16))
