"""Developer tool (GPU): host-side enqueue time vs GPU time of the training step (is the step launch-bound?).
    python tools/cpu_overhead.py [size] [batch] [steps]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
unet = importlib.import_module("3dunetcnn_amd.unet")
losses = importlib.import_module("3dunetcnn_amd.losses")
optim = importlib.import_module("3dunetcnn_amd.optim")
R = importlib.import_module("3dunetcnn_amd.synthetic")   # synthetic inputs
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().train()
crit = losses.HipDiceLoss(sigmoid=True)
opt = optim.HipAdam(m.parameters(), lr=1e-3)
x, y = R.synthetic_case(B, 4, (S, S, S))
x, y = x.cuda(), y.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    l = crit(m(x), y)
    l.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{S}^3 batch {B}: host enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, wall {1e3 * (t2 - t0) / K:.2f} ms/step")
