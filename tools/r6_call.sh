#!/bin/bash
out=gpurun_out/r6l
mkdir -p $out
root=$(pwd)
for lib in tree tools/libvar_wgrabl.so; do
python - "$lib" <<'PY'
import sys, os, importlib, torch, time
sys.path.insert(0, os.getcwd())
lib = importlib.import_module("3dunetcnn_amd._lib")
if sys.argv[1] != "tree": lib.LIB_PATH = os.path.abspath(sys.argv[1])
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
for cin, cout, s in ((32, 32, 128), (64, 64, 64), (128, 128, 32), (256, 256, 16)):
    x = be.empty_act(2, s, s, s, cin); x.buf.normal_()
    dy = be.empty_act(2, s, s, s, cout); dy.buf.normal_()
    dw = torch.empty(cout, cin, 3, 3, 3, device="cuda")
    f = lambda: be.conv_wgrad(x, dy, dw, 3, 1)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); print(sys.argv[1][-16:], cin, cout, s, f"{(time.perf_counter() - t0) / 20 * 1e3:.4f} ms (wgrad + reduce)")
PY
done
