"""Developer tool (GPU): launch time of one 3x3x3 weight gradient after a clock-ramping warm-up (tools/one_conv.py with 20 launches does
not ramp the clock).  [ONE_CONV_LIB=variant.so] python tools/bench_wgrad.py <cin> <cout> <size>"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ONE_CONV_LIB"):
    importlib.import_module("3dunetcnn_amd._lib").LIB_PATH = os.path.abspath(os.environ["ONE_CONV_LIB"])
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
cin, cout, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 2
x = be.empty_act(n, s, s, s, cin); x.buf.normal_()
dy = be.empty_act(n, s, s, s, cout); dy.buf.normal_()
dw = torch.empty(cout, cin, 3, 3, 3, device=be.device)
sc = torch.ones(n, cin, device=be.device); sh = torch.zeros(n, cin, device=be.device)
run = lambda: be.conv_wgrad(x, dy, dw, 3, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
best = 1e9
for rnd in range(3):
    for _ in range(30):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print(f"{best:.4f} ms/launch")
