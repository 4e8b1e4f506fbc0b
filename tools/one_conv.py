"""Developer tool (GPU): launch one conv layer a few times (for rocprofv3 --pmc runs).
    [ONE_CONV_LIB=tools/libvar_x.so] python tools/one_conv.py <precision> <cin> <cout> <size> [fwd|fwdplain|fwdmom|fwdnormmom|wgrad] [iters]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ONE_CONV_LIB"):          # a variant build of the library (tools/build_variant.sh) instead of the in-tree one
    importlib.import_module("3dunetcnn_amd._lib").LIB_PATH = os.path.abspath(os.environ["ONE_CONV_LIB"])
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
prec, cin, cout, s = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
what = sys.argv[5] if len(sys.argv) > 5 else "fwd"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
be.set_precision(prec)
n = 2
x = be.empty_act(n, s, s, s, cin); x.buf.normal_()
y = be.empty_act(n, s, s, s, cout); y.buf.normal_()
w = torch.randn(cout, cin, 3, 3, 3, device=be.device) * 0.05
wp = be.pack_weight(w, 0)
dw = torch.empty_like(w)
sc = torch.ones(n, cin, device=be.device); sh = torch.zeros(n, cin, device=be.device)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters):
    if it == 1:
        e0.record()
    if what == "fwd":
        be.conv_fwd(x, wp, y, 3, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
    elif what == "fwdplain":
        be.conv_fwd(x, wp, y, 3, 1)
    elif what == "fwdmom":                   # plain input, fused moments of the output
        be.conv_fwd(x, wp, y, 3, 1, moments=True)
    elif what == "fwdnormmom":
        be.conv_fwd(x, wp, y, 3, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh, moments=True)
    else:
        be.conv_wgrad(x, y, dw, 3, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
e1.record()
torch.cuda.synchronize()
if iters > 1:
    ms = e0.elapsed_time(e1) / (iters - 1)
    print(f"{' '.join(sys.argv[1:6])}: {ms:.3f} ms/launch, {2.0 * n * s ** 3 * cin * cout * 27 / ms / 1e9:.1f} TFLOP/s")
