#!/usr/bin/env python3
"""Host side of one training step, timed WITHOUT a GPU (developer tool, test infrastructure: uses the CPU-emulated twin of the library).

bench.py reports `per_rank_host_enqueue_ms_per_step`: the time a rank's Python thread needs to enqueue one step (≈600 C-ABI launches in
the eager form). With MI355_EMU_NOEXEC=1 the emulator's launch returns at once, so the same Python + ctypes + C dispatch path runs at the
real shapes on this container's CPU and can be profiled (cProfile) and A/B'd here; the GPU box then only confirms the number.
The tensors are CPU tensors (torch.empty on the CPU allocator instead of the HIP caching allocator) and nothing is computed: outputs are
garbage. Numbers are relative, for comparing host-side changes.

  python tools/host_enqueue.py [--precision fp32|bf16] [--batch 2] [--size 128] [--steps 5] [--profile]
  python tools/host_enqueue.py --gpu ...      the same loop on the real library and an idle MI355X (synchronised between steps, only the
                                              enqueue is timed): what bench.py reports, with the cProfile breakdown
"""
import argparse
import cProfile
import ctypes
import importlib
import os
import pstats
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--gpu" not in sys.argv:
    os.environ["MI355_EMU_NOEXEC"] = "1"

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--gpu", action="store_true")
    args = ap.parse_args()
    if not args.gpu:
        subprocess.check_call([os.path.join(ROOT, "tools", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    unet = importlib.import_module("3dunetcnn_amd.unet")
    losses = importlib.import_module("3dunetcnn_amd.losses")
    optim = importlib.import_module("3dunetcnn_amd.optim")
    ops = importlib.import_module("3dunetcnn_amd.ops")
    lib_mod = importlib.import_module("3dunetcnn_amd._lib")
    # the one torch op of a step that touches a logits-sized tensor (d(loss)/d(logits) times the incoming scalar: an asynchronous
    # elementwise kernel on the GPU, 100 ms of real arithmetic on CPU tensors) is skipped like the launches are
    if not args.gpu:
        losses._scaled = lambda ctx, g: ctx.dlogits
    dev = torch.device("cuda", 0) if args.gpu else torch.device("cpu")
    torch.manual_seed(0)
    model = unet.HipUNet3D(n_features=4, n_outputs=3).to(dev)
    if args.precision in ("bf16", "fp16"):
        model.act_storage = torch.bfloat16 if args.precision == "bf16" else torch.float16
    model.train()
    model.flatten_parameters()
    criterion = losses.HipDiceLoss(sigmoid=True)
    optimizer = optim.HipAdam(model.parameters(), lr=1e-3)
    if args.gpu:
        be = ops.default_backend()
    else:
        be = ops.Backend(lib=lib_mod.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "emu", "libmi355unet3d_emu.so"))), device="cpu")
        model._be = criterion._be = optimizer._be = be
    be.set_precision(args.precision)
    S = args.size
    x = torch.rand(args.batch, 4, S, S, S, device=dev) if args.gpu else torch.empty(args.batch, 4, S, S, S)
    y = (torch.rand(args.batch, 3, S, S, S, device=dev) > 0.5).float() if args.gpu else torch.empty(args.batch, 3, S, S, S)
    sync = torch.cuda.synchronize if args.gpu else (lambda: None)

    def step():
        optimizer.zero_grad(set_to_none=True)
        loss = criterion(model(x), y)
        loss.backward()
        optimizer.step()

    for _ in range(2):
        step()
    sync()
    ts = []
    for _ in range(args.steps):
        t = time.perf_counter()
        step()
        ts.append((time.perf_counter() - t) * 1e3)
        sync()
    print(f"host side of one step ({args.precision}, batch {args.batch}, {S}^3, {'MI355X, idle device' if args.gpu else 'launches skipped'}): "
          f"min {min(ts):.2f} ms, median {sorted(ts)[len(ts) // 2]:.2f} ms")
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.steps):
            step()
            pr.disable()
            sync()
            pr.enable()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(35)


if __name__ == "__main__":
    main()
