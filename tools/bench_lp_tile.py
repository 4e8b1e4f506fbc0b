"""Developer tool (GPU): the 3x3x3 stride-1 convolutions of the C3 step that run on the 16-bit TILE kernel (>= 128 input channels: no plane-ring
form), bf16 tensors, per layer shape and mode: ms per launch and TFLOP/s.   [MI355_BF16_WIDE=0] [ONE_CONV_LIB=lib.so] python tools/bench_lp_tile.py [batch]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ONE_CONV_LIB"):
    importlib.import_module("3dunetcnn_amd._lib").LIB_PATH = os.path.abspath(os.environ["ONE_CONV_LIB"])
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
be.set_precision("bf16")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
LAYERS = os.environ.get('LAYERS')
tot = 0.0
for cin, cout, s in [tuple(int(v) for v in l.split(',')) for l in LAYERS.split(';')] if LAYERS else ((128, 128, 32), (256, 128, 32), (128, 64, 64), (256, 256, 16), (512, 256, 16)):
    x = be.empty_act(n, s, s, s, cin, dtype=torch.bfloat16); x.buf.normal_()
    y = be.empty_act(n, s, s, s, cout, dtype=torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, 3, device=be.device) * 0.05
    wp = be.pack_weight(w, 0)
    sc = torch.ones(n, cin, device=be.device); sh = torch.zeros(n, cin, device=be.device)
    gx = be.empty_act(n, s, s, s, cout, dtype=torch.bfloat16); gx.buf.normal_()
    st = be.gn_stats(gx, 8, 1e-5, torch.ones(cout, device=be.device), torch.zeros(cout, device=be.device))
    for mode in ("plain", "norm+moments", "plain+gnb"):
        if mode == "plain":
            run = lambda: be.conv_fwd(x, wp, y, 3, 1)
        elif mode == "plain+gnb":
            run = lambda: be.conv_fwd(x, wp, y, 3, 1, gnb=(gx, st, 8, 0.0))
        else:
            run = lambda: be.conv_fwd(x, wp, y, 3, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh, moments=True)
        best = 1e9
        for rnd in range(3):
            for _ in range(5): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        fl = 2.0 * n * s ** 3 * cin * cout * 27
        print(f"{cin:4d} -> {cout:4d} @{s:3d}^3 x{n} {mode:13s}: {best:7.3f} ms  {fl / best / 1e9:7.1f} TFLOP/s", flush=True)
        tot += best
print(f"sum {tot:.3f} ms")
