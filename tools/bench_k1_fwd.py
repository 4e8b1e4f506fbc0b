"""Developer tool (GPU): the 1x1x1 forward / data-gradient launches of the C3 step that carry the bytes (bf16 tensors, batch 4), ms per launch and
TB/s of x + y (+ residual).   [MI355_K1_STREAM=0] [K1_FP32=1: fp32 tensors, the headline step's launches at batch 2] python tools/bench_k1_fwd.py [batch]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
DT = torch.float32 if os.environ.get("K1_FP32") else torch.bfloat16
EB = 4.0 if DT == torch.float32 else 2.0
n = int(sys.argv[1]) if len(sys.argv) > 1 else (2 if DT == torch.float32 else 4)
tot = 0.0
for cin, cout, s, res in ((64, 32, 128, False), (32, 64, 128, True), (64, 32, 64, False), (32, 64, 64, False), (64, 32, 64, True), (128, 64, 64, False)):
    x = be.empty_act(n, s, s, s, cin, dtype=DT); x.buf.normal_()
    y = be.empty_act(n, s, s, s, cout, dtype=DT)
    r = be.empty_act(n, s, s, s, cout, dtype=DT) if res else None
    if r is not None: r.buf.normal_()
    w = (torch.randn(cout, cin, 1, 1, 1, device=be.device) * 0.1).contiguous()
    wp = be.pack_weight(w, 0)
    run = lambda: be.conv_fwd(x, wp, y, 1, 1, pad=0, residual=r)
    best = 1e9
    for rnd in range(3):
        for _ in range(10): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    by = EB * n * s ** 3 * (cin + cout * (2 if res else 1))
    print(f"{cin:4d} -> {cout:4d} @{s:3d}^3 x{n}{' + residual' if res else ''}: {best:7.3f} ms  {by / best / 1e9:5.2f} TB/s", flush=True)
    tot += best
print(f"sum {tot:.3f} ms")
