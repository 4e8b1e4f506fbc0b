"""Developer tool (GPU): time the fp32 wgrad on a few layer shapes."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
n = 2
for cin, cout, s in ((32, 32, 128), (64, 32, 128), (64, 64, 64), (128, 128, 64), (256, 256, 32)):
    x = be.empty_act(n, s, s, s, cin); x.buf.normal_()
    dy = be.empty_act(n, s, s, s, cout); dy.buf.normal_()
    dw = torch.empty(cout, cin, 3, 3, 3, device=be.device)
    sc = torch.ones(n, cin, device=be.device); sh = torch.zeros(n, cin, device=be.device)
    f = lambda: be.conv_wgrad(x, dy, dw, 3, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
    for _ in range(2): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    fl = 2.0 * n * s ** 3 * cin * cout * 27
    print(f"stagger={os.environ.get('MI355_WGRAD_STAGGER', 'default')} {cin}->{cout} @{s}^3: {best:.3f} ms {fl / best / 1e9:.1f} TF/s", flush=True)
