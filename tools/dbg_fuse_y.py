"""Developer tool: is the conv OUTPUT bitwise the same with and without the fused-statistics epilogue? python tools/dbg_fuse_y.py [emu]"""
import sys, importlib, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import op_cases as C
ops = importlib.import_module("3dunetcnn_amd.ops"); lib_mod = importlib.import_module("3dunetcnn_amd._lib")
if len(sys.argv) > 1 and sys.argv[1] == "emu":
    be = ops.Backend(lib=lib_mod.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "emu", "libmi355unet3d_emu.so"))), device="cpu")
else:
    be = ops.default_backend()
def run(n, cin, cout, dhw, stride=1, norm=True, residual=False):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, *dhw, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (cin * 27) ** 0.5)
    od = [(s + 2 - 3) // stride + 1 for s in dhw]
    res = torch.randn(n, cout, *od, generator=g) if residual else None
    xa = C.to_act(be, x)
    kw = {}
    if norm:
        gi = 8 if cin % 8 == 0 else cin
        mr, sc, sh = be.gn_stats(xa, gi, 1e-5, None, None)
        kw = dict(in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
    ys = []
    for mom in (False, True, True):
        ya = C.to_act(be, torch.zeros(n, cout, *od))
        be.conv_fwd(xa, be.pack_weight(C.dev(be, wt), 0), ya, 3, stride, 1, residual=C.to_act(be, res) if residual else None, moments=mom, **kw)
        ys.append(C.from_act(ya))
    d = float((ys[0] - ys[1]).abs().max()); d2 = float((ys[1] - ys[2]).abs().max())
    print(f"n{n} {cin}->{cout} {dhw} s{stride} norm{norm} res{residual}: |y(plain) - y(fused)| = {d:.2e} ({'BITWISE' if d == 0 else 'DIFFERENT'}), fused run-to-run {d2:.2e}, |y| max {float(ys[0].abs().max()):.2f}", flush=True)
run(1, 4, 8, (16, 20, 24))
run(1, 8, 8, (16, 20, 24))
run(1, 8, 8, (16, 20, 24), residual=True)
run(1, 8, 8, (16, 20, 24), stride=2, norm=False)
run(1, 8, 16, (8, 10, 12))
run(1, 16, 16, (8, 10, 12), residual=True)
run(1, 32, 32, (16, 16, 16))
run(2, 32, 32, (64, 64, 64))
run(2, 32, 64, (64, 64, 64))
