"""Worker for tests/test_ddp_gloo.py: one rank of a world_size-2 data-parallel step.

The gradient exchange (3dunetcnn_amd/ddp.py) is exercised over gloo. Device "cpu" (default): the kernels are run by the CPU
emulator build of the same sources (tools/emu -- test infrastructure). Device "cuda" (the -m gpu variant): both ranks share
cuda:0 and run the real HIP library through the product modules (RCCL refuses two ranks on one device, gloo reduces device
tensors through the host; the reducer, its callbacks from the explicit backward and the flat device gradient buffer are the
product code either way). Each rank trains on its own synthetic batch; rank r saves its averaged gradients, updated flat
parameters and loss to <out>/rank<r>.pt.
"""
import ctypes
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet3d_ref as R  # noqa: E402  (synthetic inputs only)


def main(out_dir, device="cpu", mode="step"):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    on_gpu = device == "cuda"
    unet = importlib.import_module("3dunetcnn_amd.unet")
    losses = importlib.import_module("3dunetcnn_amd.losses")
    optim = importlib.import_module("3dunetcnn_amd.optim")
    ddp = importlib.import_module("3dunetcnn_amd.ddp")
    lib_mod = importlib.import_module("3dunetcnn_amd._lib")
    ops = importlib.import_module("3dunetcnn_amd.ops")
    if on_gpu:
        torch.cuda.set_device(0)
        be = ops.default_backend()                    # the HIP library (raises if it is missing)
    else:
        be = ops.Backend(lib=lib_mod.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "emu", "libmi355unet3d_emu.so"))), device="cpu")

    torch.manual_seed(100 + rank)                  # ranks start from DIFFERENT weights: broadcast must fix that
    kw = dict(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1, 1])
    # "lpstep": BASELINE configs[2]'s network -- bf16 operands, activations stored as bf16 (HipAutocastUNet's default) -- under the same reducer
    m = (unet.HipAutocastUNet(**kw) if mode == "lpstep" else unet.HipUNet3D(**kw)).eval()
    if on_gpu:
        m = m.cuda()
    m._be = be
    m.flatten_parameters()
    # small buckets so that the step uses several all-reduces launched from inside backward
    red = ddp.GradientBucketReducer(m, bucket_bytes=16 << 10)
    red.broadcast_parameters(0)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    crit = losses.HipDiceLoss(sigmoid=True)
    crit._be = be
    opt = optim.HipAdam(m.parameters(), lr=1e-3)
    opt._be = be
    x, y = R.synthetic_case(1, 4, (16, 16, 16), 3, seed=rank)
    if on_gpu:
        x, y = x.cuda(), y.cuda()
    rec = {"sd0": sd0, "losses": [], "n_buckets": None}
    if mode == "lpstep":
        assert m.act_storage == torch.bfloat16 and m.conv_precision == "bf16"
        opt.zero_grad(set_to_none=True)
        with red.no_sync():                                  # this rank's own gradient, not exchanged
            crit(m(x), y).backward()
        rec["local"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss = crit(m(x), y)
            loss.backward()
            if "grads" not in rec:
                rec["grads"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
                rec["n_buckets"] = len(red.buckets)
            opt.step()
            m.mark_parameters_updated()
            rec["losses"].append(float(loss))
        rec["sd2"] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        torch.save(rec, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.destroy_process_group()
        return
    if mode == "accum":
        # two micro-batches per optimizer step: the first accumulates locally (reducer.no_sync()), the second reduces the accumulated sum
        x2, y2 = R.synthetic_case(1, 4, (16, 16, 16), 3, seed=rank + 10)
        if on_gpu:
            x2, y2 = x2.cuda(), y2.cuda()
        opt.zero_grad(set_to_none=True)
        with red.no_sync():
            crit(m(x), y).backward()
        local = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        crit(m(x2), y2).backward()
        rec["local_first"] = local
        rec["grads"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        opt.step()
        rec["sd2"] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        torch.save(rec, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.destroy_process_group()
        return
    if mode == "mix4":
        # world size 4, the forms of the exchange one after the other on the same reducer, with the ranks SKEWED in time (rank r sleeps
        # before it reports gradients: collectives are launched at different moments on different ranks, buckets complete unevenly):
        #   A eager bucketed step -> every bucket launched from inside backward, exactly once
        #   B gradient accumulation: no_sync() micro-batch + closing micro-batch -> no bucket launch, one flat all-reduce of the sum
        #   C graphed step (uncaptured on the CPU emulator) -> callbacks detached, one flat all-reduce, callbacks restored
        #   D eager bucketed step again -> the per-bucket path is intact after B and C
        import time
        graph = importlib.import_module("3dunetcnn_amd.graph")
        inner = red._on_ready

        def skewed(params):
            time.sleep(0.01 * ((rank * 7) % 4))              # 0, 30, 20, 10 ms: no two ranks in step
            inner(params)
        m.grad_ready_callback = skewed
        rec["launched"] = []

        def eager():
            opt.zero_grad(set_to_none=True)
            loss = crit(m(x), y)
            loss.backward()
            rec["launched"].append((red.n_launched, len(red.buckets)))
            g = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
            opt.step()
            m.mark_parameters_updated()
            return float(loss), g
        l0, rec["grads"] = eager()                           # A
        rec["losses"].append(l0)
        rec["n_buckets"] = len(red.buckets)
        x2, y2 = R.synthetic_case(1, 4, (16, 16, 16), 3, seed=rank + 10)
        opt.zero_grad(set_to_none=True)                      # B
        with red.no_sync():
            crit(m(x), y).backward()
            rec["launched"].append((red.n_launched, 0))
        crit(m(x2), y2).backward()
        rec["launched"].append((red.n_launched, 0))
        rec["grads_accum"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        rec["sd_before_accum_step"] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        opt.step()
        m.mark_parameters_updated()
        stepper = graph.HipGraphedTrainStep(m, crit, opt, x, y, capture=False)      # C
        stepper(x, y)
        m.mark_parameters_updated()
        assert m.grad_ready_callback is skewed and m.grad_sync_callback is not None
        eager()                                              # D
        rec["sd2"] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        torch.save(rec, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.destroy_process_group()
        return
    if mode == "graphstep":
        # HipGraphedTrainStep with the reducer attached (graph.py): forward / backward without the per-bucket callbacks, ONE all-reduce
        # of the flat gradient buffer, Adam. Captured as a HIP graph on the GPU; uncaptured (the same host logic) on the CPU emulator.
        graph = importlib.import_module("3dunetcnn_amd.graph")
        stepper = graph.HipGraphedTrainStep(m, crit, opt, x, y, capture=on_gpu)
        assert stepper.reducer is red
        for i in range(2):
            loss = stepper(x, y)
            if i == 0:
                rec["grads"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
                rec["n_buckets"] = 0
            m.mark_parameters_updated()
            rec["losses"].append(float(loss))
        assert m.grad_ready_callback is not None and m.grad_sync_callback is not None      # the reducer is attached again
        rec["sd2"] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        torch.save(rec, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.destroy_process_group()
        return
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x), y)
        loss.backward()            # no reducer.wait(): the reference's loop (training_utils.py:71-72) never calls one -- the engine joins
        if "grads" not in rec:
            rec["grads"] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
            rec["n_buckets"] = len(red.buckets)
        opt.step()
        m.mark_parameters_updated()
        rec["losses"].append(float(loss))
    rec["sd2"] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    torch.save(rec, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "cpu", sys.argv[3] if len(sys.argv) > 3 else "step")
