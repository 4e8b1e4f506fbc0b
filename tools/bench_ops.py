"""Per-layer microbenchmark of the conv kernels at UNet3D's layer shapes (developer tool; needs an MI355X)."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("3dunetcnn_amd.ops")


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    be = ops.default_backend()
    n = int(os.environ.get("N", "2"))
    precs = sys.argv[1:] or ["fp32"]
    for prec in precs:
        be.set_precision(prec)
        print(f"==== precision {prec} ====")
        run(be, n, only_k3s1=(prec != "fp32"))
    be.set_precision("fp32")


def run(be, n, only_k3s1=False):
    layers = [  # (cin, cout, size, kd, stride)
        (4, 32, 128, 3, 1), (32, 32, 128, 3, 1), (64, 32, 128, 3, 1), (32, 32, 128, 3, 2),
        (32, 64, 64, 3, 1), (64, 64, 64, 3, 1), (128, 128, 64, 3, 1), (64, 64, 64, 3, 2),
        (64, 128, 32, 3, 1), (128, 128, 32, 3, 1), (256, 256, 32, 3, 1), (128, 128, 32, 3, 2),
        (128, 256, 16, 3, 1), (256, 256, 16, 3, 1),
        (64, 32, 128, 1, 1), (128, 32, 64, 1, 1), (256, 64, 32, 1, 1), (256, 128, 16, 1, 1),
    ]
    print(f"{'layer':34s} {'fwd ms':>8s} {'TF/s':>7s} {'GB/s':>7s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
    for cin, cout, s, kd, st in layers:
        if only_k3s1 and (kd != 3 or st != 1):
            continue
        so = s // st
        x = be.empty_act(n, s, s, s, cin); x.buf.normal_()
        y = be.empty_act(n, so, so, so, cout)
        dy = be.empty_act(n, so, so, so, cout); dy.buf.normal_()
        dx = be.empty_act(n, s, s, s, cin)
        w = torch.randn(cout, cin, kd, kd, kd, device=be.device) * 0.05
        wp = be.pack_weight(w, 0)
        wpd = be.pack_weight(w, 1)
        dw = torch.empty_like(w)
        sc = torch.ones(n, cin, device=be.device); sh = torch.zeros(n, cin, device=be.device)
        flops = 2.0 * n * so ** 3 * cin * cout * kd ** 3
        byts = 4.0 * (n * s ** 3 * cin + n * so ** 3 * cout + w.numel())
        tf = timeit(lambda: be.conv_fwd(x, wp, y, kd, st, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh))
        if st == 1:
            td = timeit(lambda: be.conv_fwd(dy, wpd, dx, kd, 1))
        else:
            td = timeit(lambda: be.conv_fwd(dy, wpd, dx, kd, 1, in_mode=ops.IN_ZERO_INSERT, out_dhw=(s, s, s)))
        tw = timeit(lambda: be.conv_wgrad(x, dy, dw, kd, st, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh))
        name = f"{cin}->{cout} @{s}^3 k{kd} s{st} N{n}"
        print(f"{name:34s} {tf:8.3f} {flops/tf/1e9:7.1f} {byts/tf/1e6:7.0f} | {td:8.3f} {flops/td/1e9:7.1f} | {tw:8.3f} {flops/tw/1e9:7.1f}", flush=True)
        del x, y, dy, dx


if __name__ == "__main__":
    main()
