"""Developer tool (GPU): step time of the headline configuration under the engine switches (side-stream weight gradients, fused norm
statistics). python tools/ab_step.py [size] [batch]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
unet = importlib.import_module("3dunetcnn_amd.unet"); losses = importlib.import_module("3dunetcnn_amd.losses")
optim = importlib.import_module("3dunetcnn_amd.optim"); R = importlib.import_module("3dunetcnn_amd.synthetic")
ops = importlib.import_module("3dunetcnn_amd.ops")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
x, y = R.synthetic_case(B, 4, (S, S, S)); x, y = x.cuda(), y.cuda()
be = ops.default_backend()
for rep in range(2):
    for side in (False, True):
        for fused in (False, True):
            be.fused_stats = fused
            torch.manual_seed(0)
            m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().train()
            m.backward_side_stream = side
            crit = losses.HipDiceLoss(sigmoid=True); opt = optim.HipAdam(m.parameters(), lr=1e-3)
            def step():
                opt.zero_grad(set_to_none=True)
                l = crit(m(x), y); l.backward(); opt.step()
                return l
            for _ in range(3): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
            # forward-only and backward-only split
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4):
                with torch.no_grad(): m(x)
            torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 4
            print(f"side {side!s:5} fused {fused!s:5}: {dt * 1e3:.2f} ms/step ({B / dt:.2f} vol/s), inference forward {tf * 1e3:.2f} ms", flush=True)
            del m, opt
