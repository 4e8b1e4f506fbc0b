"""Developer tool (GPU): A/B of the experimental ring-wgrad variants (csrc/conv3d_wgrad_exp.hip) against the shipped kernel.
    python tools/ab_wgrad_exp.py
variant 0 = the shipped algorithm rebuilt in the experimental file (control), 1 = 16x16x4 MFMA tiles (27 per wave),
2 = 8x8-voxel columns, 3 = both; +4 = the 16-in-flight slab reduction (4 = shipped ring kernel + new reduction). Times include the slab reduction, as bench.py's roofline figure does."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
n = 2
for cin, cout, s in ((32, 32, 128), (64, 32, 128), (128, 128, 64), (256, 256, 32)):
    x = be.empty_act(n, s, s, s, cin); x.buf.normal_()
    dy = be.empty_act(n, s, s, s, cout); dy.buf.normal_()
    sc = torch.rand(n, cin, device=be.device) + 0.5; sh = torch.randn(n, cin, device=be.device) * 0.1
    kw = dict(in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
    runs = {"shipped": lambda dw: be.conv_wgrad(x, dy, dw, 3, 1, **kw)}
    for v in (0, 1, 2, 3, 4, 5):
        runs[f"exp{v}"] = (lambda dw, v=v: be.conv_wgrad_ring_exp(x, dy, dw, v, **kw))
    outs, best = {}, {k: 1e9 for k in runs}
    for rnd in range(3):
        for k, f in runs.items():
            dw = outs.setdefault(k, torch.empty(cout, cin, 3, 3, 3, device=be.device))
            for _ in range(2): f(dw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f(dw)
            e1.record(); torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) / 5)
    fl = 2.0 * n * s ** 3 * cin * cout * 27
    ref = outs["shipped"]
    print(f"{cin}->{cout} @{s}^3: " + "  ".join(
        f"{k} {best[k]:.3f} ms {fl / best[k] / 1e9:.1f} TF/s (d {float((outs[k] - ref).abs().max() / ref.abs().max()):.1e})" for k in runs), flush=True)
