"""Developer tool (GPU): the norm-backward pass (mi355_gn_act_bwd: partial sums + finalize + apply) on the largest tensor of the step, 32 channels
@128^3; run under `rocprofv3 --kernel-trace --stats` for the per-kernel averages.   [MI355_GN_APPLY_ROWS=0] python tools/bench_gn_bwd.py [bf16] [batch]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float32
n = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 2
for c, s in ((32, 128), (64, 64), (128, 32)):
    x = be.empty_act(n, s, s, s, c, dtype=dt); x.buf.normal_()
    dA = be.empty_act(n, s, s, s, c, dtype=dt); dA.buf.normal_()
    dx = be.empty_act(n, s, s, s, c, dtype=dt)
    gamma = torch.ones(c, device=be.device); beta = torch.zeros(c, device=be.device)
    x32 = x if dt == torch.float32 else be.cast(x, torch.float32)
    mr, sc, sh = be.gn_stats(x32, 8, 1e-5, gamma, beta)
    dg, db = torch.empty(c, device=be.device), torch.empty(c, device=be.device)
    run = lambda: be.gn_act_bwd(x, dA, dx, 8, 0.0, gamma, mr, sc, sh, dg, db)
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = x.buf.element_size() * n * s ** 3 * c
    print(f"{c} ch @{s}^3 x{n} {str(dt)[6:]}: {ms:.3f} ms per gn_act_bwd (5 tensor passes = {5 * by / ms / 1e9:.2f} TB/s; the apply kernel alone: see the trace)", flush=True)
