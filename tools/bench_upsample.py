"""Developer tool (GPU): the three trilinear x2 up-sampling launches of UNet3D 128^3 (forward and backward), ms per launch.
[ONE_CONV_LIB=variant.so] python tools/bench_upsample.py [bf16] [batch]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ONE_CONV_LIB"):
    importlib.import_module("3dunetcnn_amd._lib").LIB_PATH = os.path.abspath(os.environ["ONE_CONV_LIB"])
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float32
n = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 2
tf = tb = 0.0
for c, s in ((32, 64), (64, 32), (128, 16)):
    lo = be.empty_act(n, s, s, s, c, dtype=dt); lo.buf.normal_()
    cat = be.empty_act(n, 2 * s, 2 * s, 2 * s, c, dtype=dt); cat.buf.normal_()
    def t(f):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20
    f, b = t(lambda: be.upsample2x_fwd(lo, cat, (0, 0, 0))), t(lambda: be.upsample2x_bwd(cat, lo, (0, 0, 0)))
    by = cat.buf.element_size() * n * c * (8 + 1) * s ** 3
    print(f"{c} ch {s}^3 -> {2 * s}^3 x{n} {str(dt)[6:]}: fwd {f:.3f} ms ({by / f / 1e9:.2f} TB/s)  bwd {b:.3f} ms ({by / b / 1e9:.2f} TB/s)", flush=True)
    tf += f; tb += b
print(f"sum fwd {tf:.3f} bwd {tb:.3f} ms; checksum {float(lo.buf.float().sum()):.6e}")
