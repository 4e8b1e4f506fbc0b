"""Developer tool (GPU): timings of the other BASELINE.json configurations (parity-test cases, not bench lines) for DESIGN.md.
C2 fp32 is bench.py's headline. Prints one line per configuration."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = importlib.import_module("3dunetcnn_amd.synthetic")   # synthetic inputs
unet = importlib.import_module("3dunetcnn_amd.unet"); dyn = importlib.import_module("3dunetcnn_amd.dynunet")
losses = importlib.import_module("3dunetcnn_amd.losses"); optim = importlib.import_module("3dunetcnn_amd.optim")
inferer = importlib.import_module("3dunetcnn_amd.inferer"); ops = importlib.import_module("3dunetcnn_amd.ops")
graph = importlib.import_module("3dunetcnn_amd.graph")
be = ops.default_backend()


def train_rate(model, batch, dhw, steps=4, warm=2, precision=None, graphed=False):
    model = model.cuda().train()
    model.conv_precision = precision
    crit = losses.HipDiceLoss(sigmoid=True); opt = optim.HipAdam(model.parameters(), lr=1e-3)
    x, y = R.synthetic_case(batch, 4, dhw, 3)
    x, y = x.cuda(), y.cuda()
    def step():
        opt.zero_grad(set_to_none=True)
        l = crit(model(x), y); l.backward(); opt.step()
    if graphed:      # the same step replayed as one HIP graph (3dunetcnn_amd/graph.py)
        gstep = graph.HipGraphedTrainStep(model, crit, opt, x, y)
        step = lambda: gstep(x, y)
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    return batch / dt, dt * 1e3, mem


torch.manual_seed(1234)
brats = dict(spatial_dims=3, in_channels=4, out_channels=3, kernel_size=[3] * 6, strides=[1] + [2] * 5, upsample_kernel_size=[2] * 5,
             filters=[64, 96, 128, 192, 256, 384])
rows = [
    ("C1 DynUNet (BraTS config) 1x4x64^3 fp32 train", lambda: train_rate(dyn.HipDynUNet(**brats), 1, (64, 64, 64))),
    ("C1 UNet3D 1x4x64^3 fp32 train", lambda: train_rate(unet.HipUNet3D(n_features=4, n_outputs=3), 1, (64, 64, 64))),
    ("C1 DynUNet (BraTS config) 1x4x64^3 fp32 train, HIP-graph step", lambda: train_rate(dyn.HipDynUNet(**brats), 1, (64, 64, 64), steps=8, graphed=True)),
    ("C1 UNet3D 1x4x64^3 fp32 train, HIP-graph step", lambda: train_rate(unet.HipUNet3D(n_features=4, n_outputs=3), 1, (64, 64, 64), steps=8, graphed=True)),
    ("C2 UNet3D 128^3 batch 2 fp32 train, HIP-graph step", lambda: train_rate(unet.HipUNet3D(n_features=4, n_outputs=3), 2, (128, 128, 128), graphed=True)),
    ("C2' DynUNet (BraTS config) 128^3 batch 2 fp32 train", lambda: train_rate(dyn.HipDynUNet(**brats), 2, (128, 128, 128))),
    ("C3 UNet3D 128^3 batch 4 bf16 mixed (HipAutocastUNet) train, 1 GPU", lambda: train_rate(unet.HipAutocastUNet(n_features=4, n_outputs=3), 4, (128, 128, 128), precision="bf16")),
    ("C4 UNet3D 5 levels (96.8M params) 160x192x128 batch 1 fp32 train", lambda: train_rate(unet.HipUNet3D(n_features=4, n_outputs=3, encoder_blocks=[1, 2, 2, 2, 4]), 1, (160, 192, 128))),
]
for name, f in rows:
    torch.cuda.reset_peak_memory_stats()
    r, ms, mem = f()
    print(f"{name}: {r:.2f} volumes/s, {ms:.1f} ms/step, peak {mem:.1f} GiB", flush=True)
    torch.cuda.empty_cache()
# C5: sliding-window inference over a 240x240x155 volume, 128^3 windows, overlap 0.5 -> 18 windows
m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().eval()
x = torch.randn(1, 4, 240, 240, 155).cuda()
for prec in ("fp32", "bf16"):
    m.conv_precision = prec
    inf = inferer.HipSlidingWindowInferer((128, 128, 128), sw_batch_size=2, overlap=0.5, mode="gaussian")
    inf(x, m); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): inf(x, m)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f"C5 sliding-window inference 240x240x155, 18 windows of 128^3, UNet3D convs {prec}: {dt * 1e3:.0f} ms/volume, {18 / dt:.1f} windows/s", flush=True)
