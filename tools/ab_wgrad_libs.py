"""Developer tool (GPU): the fp32 3x3x3 weight gradient (Backend.conv_wgrad) on several builds of the library in ONE process, interleaved
per shape (same box, same clock state).  python tools/ab_wgrad_libs.py lib_a.so lib_b.so ...   (variants: tools/build_variant.sh)"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_mod = importlib.import_module("3dunetcnn_amd._lib")
ops = importlib.import_module("3dunetcnn_amd.ops")
dev = torch.device("cuda", 0)
bes = [(os.path.basename(p), ops.Backend(lib=lib_mod.bind(ctypes.CDLL(os.path.abspath(p))), device=dev)) for p in sys.argv[1:]]
print("shape".ljust(22) + "".join(n.replace("libvar_", "").replace(".so", "").rjust(10) for n, _ in bes))
for cin, cout, s in ((32, 32, 128), (64, 32, 128), (64, 64, 64), (128, 128, 64), (128, 128, 32), (256, 256, 32)):
    n = 2
    x = torch.randn(n, s, s, s, cin, device=dev); dy = torch.randn(n, s, s, s, cout, device=dev)
    xa, dya = ops.Act(x), ops.Act(dy)
    dw = torch.empty(cout, cin, 3, 3, 3, device=dev)
    sc = torch.ones(n, cin, device=dev); sh = torch.zeros(n, cin, device=dev)
    best = [1e9] * len(bes)
    for rnd in range(3):
        for i, (_, be) in enumerate(bes):
            run = lambda: be.conv_wgrad(xa, dya, dw, 3, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
            for _ in range(15 if rnd else 40):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / 20)
    print(f"{cin}->{cout}@{s}^3".ljust(22) + "".join(f"{b:10.4f}" for b in best), flush=True)
