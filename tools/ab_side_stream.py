"""Developer tool (GPU): the training step with the weight-gradient kernels on a second HIP stream (HipNetBase.backward_side_stream,
experimental) against the single-stream step: gradients must be bit-identical; prints both step times.
    python tools/ab_side_stream.py [size] [batch]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
unet = importlib.import_module("3dunetcnn_amd.unet"); losses = importlib.import_module("3dunetcnn_amd.losses")
optim = importlib.import_module("3dunetcnn_amd.optim"); R = importlib.import_module("3dunetcnn_amd.synthetic")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
x, y = R.synthetic_case(B, 4, (S, S, S)); x, y = x.cuda(), y.cuda()
res = {}
for side in (False, True):
    torch.manual_seed(0)
    m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().eval()          # eval(): no Dropout3d mask, identical arithmetic in both runs
    m.backward_side_stream = side
    crit = losses.HipDiceLoss(sigmoid=True); opt = optim.HipAdam(m.parameters(), lr=1e-3)
    def step():
        opt.zero_grad(set_to_none=True)
        l = crit(m(x), y); l.backward(); opt.step()
        return l
    opt.zero_grad(set_to_none=True)
    crit(m(x), y).backward()
    torch.cuda.synchronize()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    res[side] = (g, dt)
    print(f"side stream {side}: {dt * 1e3:.2f} ms/step, {B / dt:.2f} volumes/s", flush=True)
same = torch.equal(res[False][0], res[True][0])
print("gradients bit-identical:", same, "" if same else f"(max rel diff {float((res[False][0] - res[True][0]).abs().max() / res[False][0].abs().max()):.2e})")
