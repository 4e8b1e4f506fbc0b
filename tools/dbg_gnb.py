"""Developer tool: error of the fused vs unfused norm-backward against double truth. python tools/dbg_gnb.py [emu]"""
import sys, importlib, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import op_cases as C
ops = importlib.import_module("3dunetcnn_amd.ops"); lib_mod = importlib.import_module("3dunetcnn_amd._lib")
if len(sys.argv) > 1 and sys.argv[1] == "emu":
    be = ops.Backend(lib=lib_mod.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "emu", "libmi355unet3d_emu.so"))), device="cpu")
else:
    be = ops.default_backend()
for kw in (dict(n=1, cin=8, cout=8, dhw=(16, 20, 24)), dict(n=1, cin=8, cout=8, dhw=(16, 20, 24), groups=8), dict(n=1, cin=16, cout=16, dhw=(8, 10, 12)),
           dict(n=1, cin=32, cout=32, dhw=(16, 16, 16))):
    for fused in (True, False):
        be.fused_stats = fused
        r = C.case_gn_bwd_fused(be, expect_fused=fused, **kw)
        print(kw, "fused" if fused else "plain", {k: "%.2e" % v for k, v in r.items()}, flush=True)
be.fused_stats = True
# concat statistics
for args in ((1, 8, 8, (16, 20, 24)), (2, 32, 32, (16, 16, 16))):
    print("cat", args, "%.2e" % C.case_cat_moments(be, *args))
