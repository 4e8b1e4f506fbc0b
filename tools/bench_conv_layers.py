"""Developer tool (GPU): launch time of the headline step's conv layers, one kernel at a time, for one or several builds of the
kernel library on the same box.

    [PREC=bf16] python tools/bench_conv_layers.py [lib.so[@form] ...]  (no argument: the in-tree library; "tree" names it in a list;
                                                                        "@3d" / "@2d" selects the Winograd forward kernel: MI355_WINO_FORM)

Layers: the 3x3x3 convolutions of UNet3D 4->3 at 128^3, batch 2 (forward with the norm prologue + fused moments, the plain form, and
the plain form with the norm-backward sums in its epilogue as the data gradients run it) and the 1x1x1 shortcut. PREC selects the
arithmetic (fp32 default | bf16x3 | bf16x6 | bf16). One process per library (it is loaded once per process), two rounds, best of two printed."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = tuple(os.environ["MODES"].split(",")) if os.environ.get("MODES") else ("norm+moments", "plain", "plain+gnb")      # also: norm, moments
LAYERS = [(32, 32, 128, 3), (64, 32, 128, 3), (32, 64, 128, 3), (64, 64, 64, 3), (128, 128, 64, 3), (128, 128, 32, 3), (256, 256, 32, 3), (256, 256, 16, 3),
          (32, 64, 64, 1), (128, 64, 64, 1)]
if os.environ.get("LAYERS"):                              # LAYERS=0,6: a subset by index
    LAYERS = [LAYERS[int(i)] for i in os.environ["LAYERS"].split(",")]


def one(path):
    sys.path.insert(0, ROOT)
    if "@" in path:
        path, os.environ["MI355_WINO_FORM"] = path.split("@")
    import torch
    lib = importlib.import_module("3dunetcnn_amd._lib")
    if path != "tree":
        lib.LIB_PATH = os.path.abspath(path)
    ops = importlib.import_module("3dunetcnn_amd.ops")
    be = ops.default_backend()
    be.set_precision(os.environ.get("PREC", "fp32"))
    n = 2
    out = []
    for cin, cout, s, kd in LAYERS:
        x = be.empty_act(n, s, s, s, cin); x.buf.normal_()
        y = be.empty_act(n, s, s, s, cout)
        w = torch.randn(cout, cin, kd, kd, kd, device=be.device) * 0.05
        wp = be.pack_weight(w, 0)
        sc = torch.ones(n, cin, device=be.device); sh = torch.zeros(n, cin, device=be.device)
        gx = be.empty_act(n, s, s, s, cout); gx.buf.normal_()
        st = be.gn_stats(gx, 8, 1e-5, torch.ones(cout, device=be.device), torch.zeros(cout, device=be.device))
        for mode in MODES:
            def run():
                if mode == "plain":
                    be.conv_fwd(x, wp, y, kd, 1)
                elif mode == "norm":
                    be.conv_fwd(x, wp, y, kd, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
                elif mode == "moments":
                    be.conv_fwd(x, wp, y, kd, 1, moments=(kd == 3))
                elif mode == "plain+gnb":
                    be.conv_fwd(x, wp, y, kd, 1, gnb=(gx, st, 8, 0.0))
                else:
                    be.conv_fwd(x, wp, y, kd, 1, in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh, moments=(kd == 3))
            for _ in range(8 if mode == MODES[0] else 3):      # the first mode of a layer also pays the first touch of its fresh tensors
                run()                                           # (measured: whichever mode ran first looked 5-10 % slower)
            reps = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / reps)
    print("RESULT " + " ".join(f"{v:.4f}" for v in out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        libs = sys.argv[1:] or ["tree"]
        best = {}
        for _ in range(2):
            for p in libs:
                o = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", p], capture_output=True, text=True)
                line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
                if not line:
                    print(p, "FAILED", o.stderr[-400:])
                    continue
                v = [float(t) for t in line[0].split()[1:]]
                best[p] = [min(a, b) for a, b in zip(best.get(p, v), v)]
        names = [f"{ci}->{co}@{s}^3 k{kd} {m}" for ci, co, s, kd in LAYERS for m in MODES]
        print(f"{'layer (ms / launch)':34s}" + "".join(f"{os.path.basename(p)[-14:]:>15s}" for p in libs))
        for i, nm in enumerate(names):
            print(f"{nm:34s}" + "".join(f"{best[p][i]:15.4f}" if p in best else f"{'-':>15s}" for p in libs))
        print(f"{'sum':34s}" + "".join(f"{sum(best[p]):15.4f}" if p in best else f"{'-':>15s}" for p in libs))
