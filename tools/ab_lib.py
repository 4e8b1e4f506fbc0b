"""Developer tool (GPU): step time of the headline configuration with two builds of the kernel library on the SAME box, alternately.

    python tools/ab_lib.py tools/libmi355unet3d_head.so [rounds]

The first argument is a second libmi355unet3d.so (e.g. built from the sources of HEAD, see tools/NEXT.md); the in-tree library is
the other side. Each measurement runs in its own process (the library is loaded once per process)."""
import importlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(path, size, batch):
    sys.path.insert(0, ROOT)
    import torch
    lib = importlib.import_module("3dunetcnn_amd._lib")
    if path != "tree":
        lib.LIB_PATH = os.path.abspath(path)
    unet = importlib.import_module("3dunetcnn_amd.unet")
    losses = importlib.import_module("3dunetcnn_amd.losses")
    optim = importlib.import_module("3dunetcnn_amd.optim")
    syn = importlib.import_module("3dunetcnn_amd.synthetic")
    x, y = syn.synthetic_case(batch, 4, (size, size, size))
    x, y = x.cuda(), y.cuda()
    torch.manual_seed(0)
    m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().train()
    if os.environ.get("MI355_STORAGE") == "bf16":           # with MI355_PRECISION=bf16: the 16-bit activation storage of HipAutocastUNet
        m.act_storage = torch.bfloat16
    crit = losses.HipDiceLoss(sigmoid=True)
    opt = optim.HipAdam(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        l = crit(m(x), y)
        l.backward()
        opt.step()
        return l

    for _ in range(3):
        step()
    res = []
    for side in (True, False):
        m.backward_side_stream = side
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 10 * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        with torch.no_grad():
            m(x)
    torch.cuda.synchronize()
    tf = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{path:40s} step {res[0]:.2f} ms (side stream)  {res[1]:.2f} ms (one stream)  inference forward {tf:.2f} ms", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        other = sys.argv[1]
        rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
        for _ in range(rounds):
            for p in (other, "tree"):
                subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", p, "128", "2"])
