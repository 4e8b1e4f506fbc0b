"""Developer tool (GPU): the four scalars of the adjoint identity <conv(x;w), dy> == <w, wgrad(x, dy)> at 128^3 batch 2 for the
exact-fp32 and the 6-product split-bf16 kernels (which side moves when the identity is off by ~1e-4 of the cancelled sum?)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
S, n, cin, cout = 128, 2, 32, 32
g = torch.Generator(device="cuda").manual_seed(cin * 1000 + cout)
x = be.empty_act(n, S, S, S, cin); x.buf.normal_(generator=g)
dy = be.empty_act(n, S, S, S, cout); dy.buf.normal_(generator=g)
w = torch.randn(cout, cin, 3, 3, 3, device="cuda", generator=g) * (1.0 / (27 * cin) ** 0.5)
gamma = torch.rand(cin, device="cuda", generator=g) + 0.5
beta = torch.randn(cin, device="cuda", generator=g) * 0.3
mr, sc, sh = be.gn_stats(x, 8, 1e-5, gamma, beta)
kw = dict(in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
dot = lambda a, b: float((a.double() * b.double()).sum())
res = {}
for prec in ("fp32", "bf16x6", "bf16x3"):
    be.set_precision(prec)
    y = be.empty_act(n, S, S, S, cout); dw = torch.empty_like(w)
    be.conv_fwd(x, be.pack_weight(w, 0), y, 3, 1, **kw)
    be.conv_wgrad(x, dy, dw, 3, 1, **kw)
    res[prec] = (y.tensor().clone(), dw.clone())
    print(f"{prec}: <y,dy> = {dot(y.tensor(), dy.tensor()):.6f}   <w,dw> = {dot(w, dw):.6f}   sum|y*dy| = {float((y.tensor().double() * dy.tensor().double()).abs().sum()):.3e}")
be.set_precision("fp32")
y32, dw32 = res["fp32"]
for prec in ("bf16x6", "bf16x3"):
    y6, dw6 = res[prec]
    print(f"{prec} vs fp32: max|dy| rel {float((y6 - y32).abs().max() / y32.abs().max()):.2e}  mean signed (y6-y32) {float((y6.double() - y32.double()).mean()):.3e}  "
          f"max|ddw| rel {float((dw6 - dw32).abs().max() / dw32.abs().max()):.2e}  mean signed rel dw {float(((dw6.double() - dw32.double()) / dw32.abs().max()).mean()):.3e}")
