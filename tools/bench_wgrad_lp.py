"""Developer tool (GPU): the 3x3x3 stride-1 weight gradients of the C3 step (bf16 tensors, batch 4, normalised + activated input) per layer
shape: ms per launch (with the slab reduce) and TFLOP/s.   [MI355_WGRAD_LP_TR=0] python tools/bench_wgrad_lp.py [batch]"""
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
tot = 0.0
LAYERS = os.environ.get('LAYERS')
for cin, cout, s in [tuple(int(v) for v in l.split(',')) for l in LAYERS.split(';')] if LAYERS else ((32, 32, 128), (64, 32, 128), (64, 64, 64), (128, 64, 64), (128, 128, 32), (256, 128, 32), (256, 256, 16), (512, 256, 16), (512, 512, 8)):
    x = be.empty_act(n, s, s, s, cin, dtype=torch.bfloat16); x.buf.normal_()
    dy = be.empty_act(n, s, s, s, cout, dtype=torch.bfloat16); dy.buf.normal_()
    dw = torch.empty(cout, cin, 3, 3, 3, device=be.device)
    sc = torch.ones(n, cin, device=be.device); sh = torch.zeros(n, cin, device=be.device)
    run = lambda: be.conv_wgrad(x, dy, dw, 3, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
    best = 1e9
    for rnd in range(3):
        for _ in range(10): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    fl = 2.0 * n * s ** 3 * cin * cout * 27
    print(f"{cin:4d} -> {cout:4d} @{s:3d}^3 x{n}: {best:7.3f} ms  {fl / best / 1e9:7.1f} TFLOP/s", flush=True)
    tot += best
    del x, dy
print(f"sum {tot:.3f} ms")
# the 1x1x1 weight gradients of the same step (plain input): conv3d_wgrad_k1_lp_tr against conv3d_wgrad_mfma<1, 1>; GB/s = input bytes / time
tot = 0.0
K1_DTYPE = torch.float32 if os.environ.get("K1_FP32") else torch.bfloat16
for cin, cout, s in ((64, 32, 128), (128, 64, 64), (64, 32, 64), (128, 64, 32), (32, 64, 64), (64, 128, 32)):
    x = be.empty_act(n, s, s, s, cin, dtype=K1_DTYPE); x.buf.normal_()
    dy = be.empty_act(n, s, s, s, cout, dtype=K1_DTYPE); dy.buf.normal_()
    dw = torch.empty(cout, cin, 1, 1, 1, device=be.device)
    run = lambda: be.conv_wgrad(x, dy, dw, 1, 1, pad=0)
    best = 1e9
    for rnd in range(3):
        for _ in range(10): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"1x1x1 {cin:4d} -> {cout:4d} @{s:3d}^3 x{n}: {best:7.3f} ms  {x.buf.element_size() * n * s ** 3 * (cin + cout) / best / 1e6:7.1f} GB/s", flush=True)
    tot += best
print(f"1x1x1 sum {tot:.3f} ms")
