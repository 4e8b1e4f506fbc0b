#!/usr/bin/env python
"""bench.py -- training volumes/sec of the MI355X-native 3D U-Net hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: bench.py spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one optimizer step of the reference's inner loop (unet3d/train/training_utils.py:59-72:
zero_grad -> forward -> Dice -> backward -> Adam.step) on one synthetic batch per GPU:
BASELINE.json configs[1] = UNet3D 4ch -> 3cls (default widths, 23 970 216 params), 128^3 patch, batch 2 per GPU, fp32.
Weak scaling: every rank processes its own batch; the only exchange is the bucketed RCCL gradient all-reduce
overlapped with backward (3dunetcnn_amd/ddp.py). value = world * batch * K / max-over-ranks elapsed.

Extra objects on the JSON line (tier contract):
  roofline     -- dominant kernel (by summed HIP-event time inside the timed region): algorithmic FLOPs / time vs the
                  fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md), since the 3x3x3 convs are compute-bound in fp32
                  (SURVEY.md 7.3 #1); `hbm_gbps` gives the same launches' algorithmic bytes / time for reference.
  cpu_baseline -- the oracle (CPU restatement of the reference graph, oracle/unet3d_ref.py) timed on this host for the
                  training step at N=1 on the full 128^3 patch: 1 warm-up + 3 timed steps (rank 0, N=1 runs only; ~40 s).
"""
import argparse
import importlib
import json
import os
import sys
import time

# dmabuf IPC for RCCL between the per-GPU processes (the host driver has no legacy IPC); must be set before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBPS = 8000.0


DTYPE = {"fp32": "f32", "bf16x3": "f32 (3xbf16 split-MFMA emulation)", "bf16x6": "f32 (6xbf16 split-MFMA emulation)", "bf16": "bf16 (mixed)", "fp16": "f16 (mixed)"}
ARITH = {"fp32": "3x3x3 stride-1 convs with >= 8 channels and >= 16^3 voxels: Winograd F(2x2,3x3) x direct-z on v_mfma_f32_32x32x2_f32 (fp32 products of "
                 "TRANSFORMED operands, fp32 accumulate; per-launch error vs fp64 <= 1.2e-6, the class of a direct fp32 conv); every other conv: "
                 "direct, exact fp32 products on the same instruction",
         "bf16x3": "3x3x3 stride-1 convs: fp32 operands split hi+lo bf16, 3 v_mfma_f32_32x32x16_bf16 products per MAC, fp32 accumulate "
                   "(product error <= 2^-16); everything else fp32",
         "bf16x6": "3x3x3 stride-1 convs: fp32 operands split into 3 bf16 planes, 6 bf16 MFMA products per MAC, fp32 accumulate "
                   "(product error ~2^-23, fp32-class); everything else fp32",
         "bf16": "3x3x3 stride-1 convs: operands rounded to bf16, fp32 accumulate (autocast-style mixed precision); every other conv, the norms, "
                 "the loss and the optimizer in fp32 arithmetic",
         "fp16": "3x3x3 stride-1 convs: operands rounded to IEEE fp16 (v_mfma_f32_32x32x16_f16), fp32 accumulate -- the arithmetic of the "
                 "reference's AutocastUNet under CUDA autocast; everything else fp32"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="volumes per GPU (BASELINE configs[1]: 2)")
    ap.add_argument("--size", type=int, default=128, help="cubic patch edge (BASELINE configs[1]: 128)")
    ap.add_argument("--storage", default=None, choices=["fp32", "bf16", "fp16"],
                    help="activation storage type (HipAutocastUNet(activation_storage=...)): default bf16 with --precision bf16 on a UNet3D "
                         "(conv outputs, block outputs, concat buffers and their gradients live in HBM as bf16, as under the reference's "
                         "autocast), fp32 otherwise")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16x6", "bf16", "fp16"],
                    help="arithmetic of the 3x3x3 stride-1 convs: exact f32 MFMA (default) | split-bf16 fp32 emulation | bf16")
    ap.add_argument("--model", default="unet3d", choices=["unet3d", "dynunet"])
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configuration: c2 = configs[1], the headline `metric` is quoted on (default: UNet3D 128^3 batch 2 fp32); "
                         "c3 = configs[2] per-GPU shape (bf16 mixed, batch 4); c4 = configs[3] (5-level UNet3D, 160x192x128, batch 1); "
                         "c5 = configs[4] (sliding-window inference over 240x240x155, 128^3 windows). c3-c5 are reported for completeness "
                         "(parity-test cases, tests/test_fullsize_configs_gpu.py); the driver's line is c2")
    ap.add_argument("--graph", action="store_true",
                    help="the step as ONE replayed HIP graph (3dunetcnn_amd/graph.py): forward + loss + backward captured; with N > 1 the "
                         "gradients are exchanged in one all-reduce of the flat buffer after the replay instead of per bucket from inside "
                         "backward. ~600 host launches per step become 3 (many ranks on one host, short steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="disable the per-launch HIP events (roofline -> null)")
    ap.add_argument("--no-c3", action="store_true",
                    help="skip the short run of BASELINE configs[2]'s per-GPU shape (bf16 mixed, batch 4, 16-bit activation storage) that the "
                         "default N=1 line carries under `c3` (a child process: python bench.py --config c3 --steps 20 --warmup 5)")
    ap.add_argument("--no-precision-modes", action="store_true",
                    help="skip the short extra runs of the opt-in conv arithmetic modes (reported under precision_modes, N=1 only)")
    # TEST INFRASTRUCTURE (tests/test_bench_contract.py): run the launch / rank / reduce plumbing of this script on the CPU emulator
    # build of the kernel sources over gloo. Never a measurement: the line it prints says so in "data".
    ap.add_argument("--emulator-plumbing-test", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, the same command line, rendezvous
    on 127.0.0.1:<free port>) and wait for them. Rank 0 inherits stdout and prints the JSON line; a failing rank takes the job
    down (the others are terminated by PID). The reference turns multi-GPU on with one call (unet3d/models/build.py:18-20);
    this is the one command that does it here."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    q.terminate()
        time.sleep(0.05)
    return rc


# HBM-traffic summaries (tools/pmc_summary.py) of the rocprofv3 PMC passes of THIS command, per (config, precision)
PMC_FILES = {("c2", "fp32"): os.path.join("profiles", "r6_bench_fp32_hbm_traffic_pmc.csv"),
             ("c2", "bf16"): os.path.join("profiles", "r6_bf16_hbm_traffic_pmc.csv"),
             ("c3", "bf16"): os.path.join("profiles", "r6_c3_hbm_traffic_pmc.csv")}
# kernel family (ops.Backend prof name) -> substrings of the trace names of its instantiations
PMC_FAMILY = {"conv3d_wino2d": ("conv3d_wino2d",), "conv3d_wino3d": ("conv3d_wino3d",), "conv3d_c4_bwd (+reduce)": ("conv3d_c4_bwd",), "conv3d_k3_bf16<...>": ("conv3d_k3_bf16", "conv3d_k3_lp_zring"),
              "conv3d_wgrad_wino_ring (+reduce)": ("conv3d_wgrad_wino_ring",), "conv3d_wgrad_k3_bf16<...> (+reduce)": ("conv3d_wgrad_k3_bf16", "conv3d_wgrad_lp_ring"),
              "conv3d_wgrad_ring (+reduce)": ("conv3d_wgrad_ring",)}


def pmc_rows(config, precision):
    """Rows of the committed PMC summary of this command: [(kernel trace name, calls, HBM bytes per launch)], or (None, reason).
    FETCH_SIZE and WRITE_SIZE come from separate rocprofv3 passes; FETCH_SIZE is doubled per the gfx950 note in MI355X_MICROARCH.md
    'HBM'. PMC collection needs rocprofv3 around the process, so bench.py reads the summary -- but only one collected from THESE
    kernels: the file's first line records the sha256 of the kernel sources it was measured on (tools/pmc_summary.py), and a file
    from other sources is refused (traffic -> null, the reason in traffic_source)."""
    rel = PMC_FILES.get((config, precision))
    if rel is None or not os.path.exists(os.path.join(ROOT, rel)):
        return None, None if rel is None else f"{rel} not collected"
    import csv
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_summary import kernel_source_hash
    lines = open(os.path.join(ROOT, rel)).read().splitlines()
    recorded = None
    if lines and lines[0].startswith("#"):
        for tok in lines[0].split():
            if tok.startswith("kernel_source_sha256="):
                recorded = tok.split("=", 1)[1]
        lines = lines[1:]
    if recorded != kernel_source_hash():
        return None, f"{rel} is stale (collected from kernel sources {str(recorded)[:12]}, tree has {kernel_source_hash()[:12]}): re-run tools/gpu_profile.sh <tag> pmc"
    rows = [(r["Kernel"], int(r["Calls"]), (float(r["fetch_x2_MiB_per_launch"]) + float(r["WRITE_SIZE_MiB_per_launch"])) * 1048576.0)
            for r in csv.DictReader(lines)]
    return rows, f"{rel} (FETCH_SIZE x2 + WRITE_SIZE per launch; family figure = call-weighted over the file's rows of the family)"


def family_clock(config, precision, family):
    """Call-weighted shader clock (GHz) of `family`'s launches from the same PMC summary (column shader_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs /
    duration, a separate rocprofv3 pass), or None when the summary has no such column / is stale."""
    rows, _ = pmc_rows(config, precision)
    if rows is None:
        return None
    import csv
    lines = open(os.path.join(ROOT, PMC_FILES[(config, precision)])).read().splitlines()[1:]
    keys = PMC_FAMILY.get(family, (family.split(" (")[0].split("<")[0],))
    hit = [(int(r["Calls"]), float(r["shader_clock_ghz"])) for r in csv.DictReader(lines)
           if r.get("shader_clock_ghz") not in (None, "", "nan") and any(k in r["Kernel"] for k in keys)]
    calls = sum(c for c, _ in hit)
    return round(sum(c * g for c, g in hit) / calls, 3) if calls else None


def family_traffic(rows, family):
    """Call-weighted HBM bytes per launch over every instantiation of `family` in the PMC summary (the summary is of the same command:
    its call mix is this step's)."""
    keys = PMC_FAMILY.get(family, (family.split(" (")[0].split("<")[0],))
    hit = [(k, c, b) for k, c, b in rows if any(key in k for key in keys)]
    calls = sum(c for _, c, _ in hit)
    return (sum(c * b for _, c, b in hit) / calls, hit) if calls else (None, [])


def cpu_baseline(size, model="unet3d"):
    """The oracle graph's training step on the host CPU, SURVEY 8(d) protocol: N = 1, the FULL patch (C2's shape at N = 1), 1 warm-up +
    3 timed iterations of zero_grad -> forward -> Dice -> backward -> Adam.step with time.perf_counter, forward / backward / optimizer
    split reported. kind "port": oracle/unet3d_ref.py is a restatement of the reference graph that is bit-identical to the imported
    reference UNet3D (tests/test_oracle_pinned.py; /root/reference does not exist on the GPU box); for --model dynunet the torch
    restatement of MONAI's DynUNet in the BraTS configuration (oracle/dynunet_ref.py, unpinned). ~40 s of host time at 128^3.
    Only if the host cannot hold the full patch (MemoryError / allocator failure) is the half-edge patch timed instead; the line then
    carries "extrapolated": true, its forward / backward times are scaled by the voxel ratio and Adam (parameter-sized) is not."""
    from oracle import torch_ops as O
    cores = min(os.cpu_count() or 1, 64)     # oneDNN's conv3d does not scale past a few dozen threads on these shapes
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    if model == "dynunet":
        from oracle import dynunet_ref as DR
        dyn = importlib.import_module("3dunetcnn_amd.dynunet")
        holder = dyn.HipDynUNet(spatial_dims=3, in_channels=4, out_channels=3, kernel_size=[3] * 6, strides=[1] + [2] * 5,
                                upsample_kernel_size=[2] * 5, filters=[64, 96, 128, 192, 256, 384])
        fwd = lambda sd, x: DR.dynunet_forward(sd, x, 6)
        graph, kind = "oracle/dynunet_ref.py (torch restatement of MONAI DynUNet, BraTS configuration; unpinned)", "port"
    else:
        from oracle import unet3d_ref as R
        unet = importlib.import_module("3dunetcnn_amd.unet")
        holder = unet.HipUNet3D(n_features=4, n_outputs=3)          # parameter container only (never run on CPU)
        fwd = lambda sd, x: R.unet3d_forward(sd, x)
        graph, kind = "oracle/unet3d_ref.py (bit-identical to the imported reference UNet3D, tests/test_oracle_pinned.py)", "port"
    from oracle import unet3d_ref as R
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in holder.state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3)

    def step(dhw):
        x, y = R.synthetic_case(1, 4, dhw)
        opt.zero_grad()
        t0 = time.perf_counter()
        out = fwd(sd, x)
        loss = O.dice_loss(out, y)
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    smallest = 64 if model == "dynunet" else 32                  # five stride-2 levels need >= 64^3 (InstanceNorm over > 1 voxel)
    step((smallest,) * 3)                                        # thread pools, oneDNN primitives
    edge, mult = size, 1.0
    try:
        step((edge,) * 3)                                        # the warm-up iteration of the protocol
        runs = [step((edge,) * 3) for _ in range(3)]
    except (MemoryError, RuntimeError) as e:                     # host RAM too small for the full patch: the only case that is scaled
        # torch reports a failed host allocation as a RuntimeError ("DefaultCPUAllocator: not enough memory", "can't allocate memory");
        # any OTHER RuntimeError is a bug in the baseline and must not turn into a silent half-size run
        if not isinstance(e, MemoryError) and not any(t in str(e).lower() for t in ("not enough memory", "allocate memory", "out of memory")):
            raise
        edge = max(smallest, size // 2)
        mult = (size / edge) ** 3
        step((edge,) * 3)
        runs = [step((edge,) * 3) for _ in range(3)]
    f, b = (sum(r[i] for r in runs) / 3 * mult for i in range(2))
    o = sum(r[2] for r in runs) / 3                              # Adam walks the parameters, not the voxels: never scaled
    tot = f + b + o
    out = {"value": round(1.0 / tot, 5), "unit": "volumes/s", "cores": cores, "kind": kind,
           "sample": f"1 warm-up + 3 timed training steps (zero_grad -> forward -> sigmoid-Dice -> backward -> Adam.step) of {graph}, N=1, "
                     f"{edge}^3 patch, fp32, {cores} threads: {tot:.2f} s/step",
           "seconds_per_step": round(tot, 3), "forward_s": round(f, 3), "backward_s": round(b, 3), "optimizer_s": round(o, 3),
           "per_iteration_s": [round((r[0] + r[1]) * mult + r[2], 3) for r in runs], "extrapolated": mult != 1.0}
    if mult != 1.0:
        out["sample"] += f" -- EXTRAPOLATED: the host could not hold the {size}^3 step; forward / backward of the {edge}^3 patch x {mult:.0f}"
    return out


C3_KEYS = ("volumes_per_s_per_gpu", "ms_per_step", "workload", "activation_storage", "dtype", "steps", "warmup", "bound", "frac", "kernel",
           "mfma_pipe_frac", "hbm_frac", "traffic_over_algorithmic", "share_of_step", "per_rank_host_enqueue_ms_per_step", "command")


def c3_block(steps=20, warmup=5, timeout=600):
    """BASELINE configs[2] per GPU (the same model in bf16 mixed precision, batch 4 per GPU, 16-bit activation storage: the configuration
    the 8-GPU DDP run is quoted on, reference: AutocastUNet, segmentation/unet.py:53-58) measured by a CHILD process on this GPU after the
    headline's timed region: `python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline` (5 + 2 steps were measured 16 % slow: the
    first steps of a fresh process run before the allocator and the clocks have settled, profiles/r5_c3_warmup.txt). A separate process because the step
    allocates the batch-4 activations of another network; its own line is what `--config c3` prints, this block keeps the figures a
    reader of the driver's line needs. Informational: never `value`."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "c3", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"child exited {r.returncode}: {(r.stderr or r.stdout)[-300:]}", "command": " ".join(cmd[1:])}
        c = json.loads(line[-1])
    except Exception as e:                                       # a failing extra must not take the headline line down; it says so
        return {"error": f"{type(e).__name__}: {e}", "command": " ".join(cmd[1:])}
    roof = c.get("roofline") or {}
    return {"volumes_per_s_per_gpu": c["value"], "ms_per_step": c["ms_per_step"], "workload": c["config"]["workload"],
            "activation_storage": c["config"]["activation_storage"], "dtype": c["dtype"], "steps": c["steps"], "warmup": c["warmup"],
            "bound": roof.get("bound"), "frac": roof.get("frac"), "kernel": roof.get("kernel"),
            "mfma_pipe_frac": roof.get("mfma_pipe_frac"), "hbm_frac": roof.get("hbm_frac"),
            "traffic_over_algorithmic": roof.get("traffic_over_algorithmic"), "share_of_step": roof.get("share_of_step"),
            "per_rank_host_enqueue_ms_per_step": c.get("per_rank_host_enqueue_ms_per_step"),
            "command": "python bench.py --config c3 --steps %d --warmup %d --no-cpu-baseline (child process, same GPU, after the timed region)" % (steps, warmup)}


def bench_c5(args, unet, inferer_mod, dev):
    """BASELINE configs[4]: sliding-window inference (unet3d/predict/volumetric.py:131-177) over one 240 x 240 x 155 volume with 128^3
    windows at overlap 0.5 (18 windows, Gaussian importance map); a "step" = one whole volume, input resident in HBM."""
    torch.manual_seed(1234)
    m = unet.HipUNet3D(n_features=4, n_outputs=3).to(dev).eval()
    m.conv_precision = args.precision
    x = torch.randn(1, 4, 240, 240, 155, generator=torch.Generator().manual_seed(0)).to(dev)
    inf = inferer_mod.HipSlidingWindowInferer((128, 128, 128), sw_batch_size=2, overlap=0.5, mode="gaussian")
    with torch.no_grad():
        for _ in range(max(1, args.warmup)):
            inf(x, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            inf(x, m)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    return {"metric": "sliding-window inference volumes/sec (240x240x155, 128^3 windows)", "value": round(1.0 / dt, 4), "unit": "volumes/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[args.precision], "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: UNet3D 4ch->3cls (23970216 params), sliding-window inference over 1x4x240x240x155, "
                                   "18 overlapping 128^3 windows (overlap 0.5, Gaussian map), 2 windows per forward",
                       "conv_arithmetic": ARITH[args.precision], "global_batch": 1, "parallelism": "dp1"},
            "roofline": None, "windows_per_s": round(18.0 / dt, 2)}


def main():
    args = parse()
    if args.config == "c3":
        args.precision, args.batch = "bf16", 4
    elif args.config == "c4":
        args.batch = 1
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))                          # no launcher: this process only starts and reaps the ranks
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; reporting n_gpus={world}", file=sys.stderr)
    emu = args.emulator_plumbing_test
    torch.set_num_threads(max(1, (os.cpu_count() or 1) // max(world, 1)))      # host-side torch ops: no oversubscription across ranks
    import torch.distributed as dist
    if emu:
        dev = torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        assert local_rank < torch.cuda.device_count(), f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    unet = importlib.import_module("3dunetcnn_amd.unet")
    losses = importlib.import_module("3dunetcnn_amd.losses")
    optim = importlib.import_module("3dunetcnn_amd.optim")
    ddp = importlib.import_module("3dunetcnn_amd.ddp")
    ops = importlib.import_module("3dunetcnn_amd.ops")
    synthetic = importlib.import_module("3dunetcnn_amd.synthetic")

    if args.config == "c5":
        if rank == 0:
            print(json.dumps(bench_c5(args, unet, importlib.import_module("3dunetcnn_amd.inferer"), dev)), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    torch.manual_seed(1234)
    if args.config == "c4":
        model = unet.HipUNet3D(n_features=4, n_outputs=3, encoder_blocks=[1, 2, 2, 2, 4]).to(dev)
        model_desc = "deep UNet3D 4ch->3cls, encoder_blocks [1,2,2,2,4] (5 levels, 96822184 params)"
    elif args.model == "dynunet":
        dyn = importlib.import_module("3dunetcnn_amd.dynunet")
        model = dyn.HipDynUNet(spatial_dims=3, in_channels=4, out_channels=3, kernel_size=[3] * 6, strides=[1] + [2] * 5,
                               upsample_kernel_size=[2] * 5, filters=[64, 96, 128, 192, 256, 384]).to(dev)   # brats2020_config.json:2-107
        model_desc = "BraTS2020-config DynUNet 4ch->3cls (24928451 params)"
    elif emu:
        model = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1, 1])   # the emulator runs ~1e4x slower
        model_desc = "REDUCED UNet3D (base_width 8, 3 levels) for the emulator plumbing test"
    else:
        model = unet.HipUNet3D(n_features=4, n_outputs=3).to(dev)
        model_desc = "UNet3D 4ch->3cls (23970216 params)"
    if args.storage is None:
        args.storage = args.precision if (args.precision in ("bf16", "fp16") and args.model != "dynunet") else "fp32"
    if args.storage in ("bf16", "fp16"):
        if args.precision != args.storage or args.model == "dynunet":
            raise SystemExit(f"--storage {args.storage} goes with --precision {args.storage} on a UNet3D")
        model.act_storage = torch.bfloat16 if args.storage == "bf16" else torch.float16
    model.train()                                                 # Dropout3d active, as in the reference's training loop
    model.flatten_parameters()
    criterion = losses.HipDiceLoss(sigmoid=True)
    optimizer = optim.HipAdam(model.parameters(), lr=1e-3)
    if emu:
        import ctypes
        lib_mod = importlib.import_module("3dunetcnn_amd._lib")
        be = ops.Backend(lib=lib_mod.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "emu", "libmi355unet3d_emu.so"))), device="cpu")
        model._be = criterion._be = optimizer._be = be
        args.no_kernel_events = args.no_cpu_baseline = args.no_precision_modes = True
    else:
        be = ops.default_backend()
    reducer = None
    if world > 1:
        reducer = ddp.GradientBucketReducer(model)
        reducer.broadcast_parameters(0)

    S, B = args.size, args.batch
    dhw = (160, 192, 128) if args.config == "c4" else (S, S, S)
    x, y = synthetic.synthetic_case(B, 4, dhw, seed=rank)
    x, y = x.to(dev), y.to(dev)                                   # inputs resident in HBM before the timed region
    be.set_precision(args.precision)

    host_s = [0.0, 0]                                             # host time spent enqueueing steps (no device sync inside), step count

    # fp16 activation storage = the reference's amp form: gradients stored as fp16 need its loss scale (train/train.py:33-37,
    # training_utils.py:60-69: scaler.scale(loss).backward(); scaler.step(optimizer)). Here GradScaler's initial scale, 2^16, held fixed
    # with the unscale fused into the Adam launch (HipAdam.grad_scale): the scaler's per-step inf check is a host synchronisation,
    # which a throughput line should not time (the GradScaler protocol itself: tests/test_boundary.py)
    scaler = 65536.0 if args.storage == "fp16" else None

    def eager_step():
        optimizer.zero_grad(set_to_none=True)
        out = model(x)
        loss = criterion(out, y)
        if scaler is not None:
            (loss * scaler).backward()
            optimizer.grad_scale = 1.0 / scaler
            optimizer.step()
            return loss
        loss.backward()                                           # the reducer's bucket all-reduces are launched from inside backward
        optimizer.step()                                          # and joined at its end (engine.py: grad_sync_callback)
        return loss

    graphed = None
    if args.graph and scaler is not None:
        raise SystemExit("--graph with fp16 activation storage: the captured step has no loss scale; use --storage fp32 or the eager step")
    if args.graph:
        graph_mod = importlib.import_module("3dunetcnn_amd.graph")
        # one flat all-reduce per step when N > 1. On the CPU emulator (plumbing test) the same step runs uncaptured: the host logic of
        # the graphed form -- callbacks detached, one exchange of the flat buffer between backward and Adam -- without a HIP graph
        graphed = graph_mod.HipGraphedTrainStep(model, criterion, optimizer, x, y, capture=dev.type == "cuda")
        args.no_kernel_events = True                              # a replayed graph has no per-launch events

    def step():
        t = time.perf_counter()
        loss = graphed(x, y) if graphed is not None else eager_step()
        host_s[0] += time.perf_counter() - t
        host_s[1] += 1
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # Per-launch HIP events go into the timed region only when its launches are serialized (MI355_SIDE_STREAM=0: the rocprofv3 command
    # of profiles/): with the weight gradients on a second stream (the default) a launch's duration is not its own, the roofline is
    # measured on extra steps after the timed region (below), and ~1200 event records per step inside the timed region are pure
    # perturbation (bf16, batch 4: 57.6 -> 75.0 ms/step with them).
    side = bool(getattr(model, "backward_side_stream", False)) and dev.type == "cuda"
    if not args.no_kernel_events and not side:
        be.prof = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    prof_timed, be.prof = be.prof, None
    loss_val = float(loss.item())
    # Host enqueue time of one step, measured OUTSIDE the timed region on an idle device: inside it a GPU-bound loop runs the host into
    # the launch queue's depth limit, and the time spent blocked there says nothing about the Python / launch work of a step.
    host_s[0], host_s[1] = 0.0, 0
    for _ in range(2):
        step()
        if dev.type == "cuda":
            torch.cuda.synchronize()
    host_ms = host_s[0] / max(host_s[1], 1) * 1e3
    # Roofline pass. The timed region runs the weight-gradient kernels on a second stream (engine.py: backward_side_stream), where a
    # launch shares the chip with the dgrad chain and its HIP-event duration is not its own. The kernel's rate is therefore measured
    # on ROOF_STEPS extra steps right after the timed region with that overlap switched off (one stream, every launch alone);
    # `value` / `ms_per_step` never include these steps. MI355_SIDE_STREAM=0 makes the timed region itself serialized (that is the
    # command the committed rocprofv3 summary in profiles/ was taken with).
    prof = prof_timed
    roof_note = "HIP events over the timed region"
    ROOF_STEPS = 3
    if not args.no_kernel_events and side:
        model.backward_side_stream = False
        step()
        barrier()
        be.prof = []
        t1 = time.perf_counter()
        for _ in range(ROOF_STEPS):
            step()
        barrier()
        serial_ms = (time.perf_counter() - t1) / ROOF_STEPS * 1e3
        prof, be.prof = be.prof, None
        model.backward_side_stream = True
        roof_note = (f"HIP events over {ROOF_STEPS} serialized steps run right after the timed region (weight gradients on the launch stream: "
                     f"{serial_ms:.2f} ms/step; the timed region overlaps them with the dgrad chain on a second stream, where a launch's "
                     "duration is not its own)")

    per_rank, per_rank_host = [dt], [host_ms]
    if world > 1:
        mine = torch.tensor([dt, host_ms], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t[0].item()) for t in every]
        per_rank_host = [float(t[1].item()) for t in every]
    dt = max(per_rank)                                            # the job is as slow as its slowest rank
    # proof of the N ranks for the driver's SCALE line: the world size the RCCL communicator reports and every rank's own device
    mine_dev = (f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)} pci bus {getattr(torch.cuda.get_device_properties(local_rank), 'pci_bus_id', -1)}"
                if dev.type == "cuda" else f"rank {rank}: cpu (emulator)")
    rank_devices, comm_world, comm_backend = [mine_dev], 1, None
    if world > 1:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine_dev)
        comm_world, comm_backend = dist.get_world_size(), dist.get_backend()

    roofline = None
    if prof:
        roof_steps = ROOF_STEPS if prof is not prof_timed else args.steps
        agg, inst = {}, {}
        for name, fl, by, e0, e1, variant, fused in prof:
            secs = e0.elapsed_time(e1) * 1e-3
            a = agg.setdefault(name, [0.0, 0.0, 0.0, 0, 0.0])
            a[0] += secs
            a[1] += fl
            a[2] += by
            a[3] += 1
            a[4] += fused
            v = inst.setdefault((name, variant or name), [0.0, 0.0, 0.0, 0, 0.0])
            v[0] += secs
            v[1] += fl
            v[2] += by
            v[3] += 1
            v[4] += fused
        name, (secs, fl, by, cnt, fused) = max(agg.items(), key=lambda kv: kv[1][0])
        ach = fl / secs / 1e12
        on_bf16 = "bf16" in name
        peak = BF16_MFMA_PEAK_TFLOPS if on_bf16 else FP32_MFMA_PEAK_TFLOPS
        products = {"bf16x3": 3, "bf16x6": 6, "bf16": 1, "fp16": 1}.get(args.precision, 1) if on_bf16 else 1
        executed = getattr(be, "WINO_EXECUTED", {}).get(name, 1.0)
        # ALGORITHMIC bytes of a launch = SURVEY 8(d)'s unfused-compulsory figure (inputs + outputs + weights, each once) PLUS the reads
        # this launch fuses on top of them (the residual; the normalised tensor of the norm-backward sums): what the kernel must move.
        alg_bytes = (by + fused) / cnt
        mfma_frac = ach * products * executed / peak
        hbm_gbps = (by + fused) / secs / 1e9
        hbm_frac = hbm_gbps / HBM_PEAK_GBPS
        # HBM traffic of the SAME launches: call-weighted over every instantiation of the family in the PMC summary of this command
        rows, traffic_src = pmc_rows(args.config, args.precision)
        if rows is not None and args.precision == "bf16" and args.storage != "bf16":
            rows, traffic_src = None, "the committed bf16 PMC summaries were collected with the default 16-bit activation storage"
        traffic, inst_rows, hit = None, [], []
        if rows is not None:
            traffic, hit = family_traffic(rows, name)
        for (fam, variant), v in sorted(inst.items(), key=lambda kv: -kv[1][0]):
            if fam != name:
                continue
            row = {"kernel": variant, "launches_per_step": v[3] // roof_steps, "avg_launch_ms": round(v[0] / v[3] * 1e3, 4),
                   "algorithmic_tflops": round(v[1] / v[0] / 1e12, 1),
                   "algorithmic_bytes_per_launch": round((v[2] + v[4]) / v[3]), "of_which_fused_reads": round(v[4] / v[3])}
            m = [(k, c, t) for k, c, t in hit if variant in k]
            if len(m) == 1:
                row["traffic_bytes_per_launch"] = round(m[0][2])
                row["traffic_over_algorithmic"] = round(m[0][2] / ((v[2] + v[4]) / v[3]), 3)
            inst_rows.append(row)
        # the binding roofline: the one that needs MORE time for this launch mix. `achieved` / `peak` / `frac` are quoted on it; both
        # fractions are always reported (mfma_pipe_frac, hbm_frac)
        bound = "hbm" if hbm_frac > mfma_frac else "mfma"
        if bound == "mfma":
            # `achieved` is ALGORITHMIC (2 * N * V * Cin * Cout * 27 per launch / duration). A Winograd kernel executes only
            # `executed` of those multiplications, so its roofline in algorithmic units is the matrix peak / executed (12/27 forward /
            # dgrad, 16/36 weight gradient): `peak` is that figure, `frac` = achieved / peak = executed rate / MFMA peak, always <= 1.
            head = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak / executed / products, 1), "unit": "TFLOP/s",
                    "frac": round(mfma_frac, 4)}
        else:
            head = {"bound": "hbm", "achieved": round(hbm_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hbm_frac, 4)}
        roofline = dict(head, kernel=name, measured=roof_note, mfma_peak_tflops=peak,
                        peak_note=(("algorithmic-equivalent peak = MFMA peak %.1f / %.4f executed-over-algorithmic multiplications" % (peak, executed))
                                   if executed != 1.0 and bound == "mfma" else
                                   "dense MFMA peak of the arithmetic type (MI355X_MICROARCH.md)" if bound == "mfma" else "HBM3E peak (MI355X_MICROARCH.md)"),
                        traffic=None if traffic is None else round(traffic), traffic_unit="bytes/launch (HBM, PMC; call-weighted over the family's launches)",
                        traffic_source=traffic_src, algorithmic_bytes_per_launch=round(alg_bytes),
                        algorithmic_bytes_note="inputs + outputs + weights once (SURVEY 8d) + the reads the launch fuses (residual, normalised tensor "
                                               "of the norm-backward sums), averaged over the same launches as `traffic`",
                        unfused_compulsory_bytes_per_launch=round(by / cnt),
                        traffic_over_algorithmic=None if traffic is None else round(traffic / alg_bytes, 3),
                        # the clock the family's launches ran at under the profiler (power-limited: below the 2.4 GHz the peak is quoted at)
                        shader_clock_ghz=family_clock(args.config, args.precision, name), shader_clock_note="GRBM_GUI_ACTIVE / 8 XCDs / duration, "
                        "rocprofv3 pass of the same serialized command (profiled runs clock a few per cent below un-profiled ones); peak figures assume 2.4 GHz",
                        instantiations=inst_rows,
                        mfma_products_per_mac=products, mfma_pipe_frac=round(mfma_frac, 4),
                        # Winograd kernels execute fewer multiplications than the algorithmic count `achieved` is quoted in (it can
                        # exceed the matrix peak): the executed rate is what the MFMA pipe sees
                        executed_over_algorithmic=round(executed, 4), algorithmic_tflops=round(ach, 2), executed_tflops=round(ach * executed, 2),
                        launches=cnt, launches_per_step=cnt // roof_steps, avg_launch_ms=round(secs / cnt * 1e3, 4),
                        hbm_gbps_algorithmic=round(hbm_gbps, 1), hbm_frac=round(hbm_frac, 4),
                        share_of_step=round(secs / roof_steps / (dt / args.steps), 4),
                        all_kernels={k: {"s": round(v[0], 5), "tflops": round(v[1] / v[0] / 1e12, 2), "hbm_gbps": round((v[2] + v[4]) / v[0] / 1e9, 1),
                                         "launches": v[3]}
                                     for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])})

    # The north star's literal conv line: the 3x3x3 conv on the 4-channel 128^3 input (first layer), forward + backward against the HBM
    # roofline. Algorithmic bytes per pass = x + y (+ weights) once: forward 604 MB, weight gradient 604 MB, data gradient 604 MB at N = 2.
    first_layer = None
    if prof:
        fl = {k: v for k, v in agg.items() if k.startswith("conv3d_c4")}
        if len(fl) >= 2:      # forward + the fused backward (conv3d_c4_bwd; round 5 and MI355_C4_BWD=0: weight gradient and data gradient apart)
            secs_f = sum(v[0] for v in fl.values())
            by_f = sum(v[2] for v in fl.values())
            first_layer = {"kernels_ms_per_launch": {k: round(v[0] / v[3] * 1e3, 4) for k, v in fl.items()},
                           "fwd_plus_bwd_ms": round(sum(v[0] / v[3] for v in fl.values()) * 1e3, 4),
                           "algorithmic_bytes_fwd_plus_bwd": round(sum(v[2] / v[3] for v in fl.values())),
                           "hbm_gbps_algorithmic": round(by_f / secs_f / 1e9, 1), "hbm_frac": round(by_f / secs_f / 1e9 / HBM_PEAK_GBPS, 4),
                           "arithmetic": ARITH[args.precision],
                           "bound": "exact fp32: 29 GFLOP per pass on the fp32 matrix / packed vector pipes = >= 0.18 ms per pass against 0.08-0.12 ms "
                                    "of HBM time, i.e. the layer is arithmetic-bound in the headline precision and 60 % of 8 TB/s is out of "
                                    "reach there (DESIGN.md section 3); the 16-bit modes move the forward to the bf16 pipe"}
    if rank == 0:
        out = {"metric": (f"training volumes/sec ({dhw[0]}^3, 4ch->3cls)" if dhw[0] == dhw[1] == dhw[2] else
                          f"training volumes/sec ({'x'.join(str(v) for v in dhw)}, 4ch->3cls{', 5 levels' if args.config == 'c4' else ''})"),
               "value": round(world * B * args.steps / dt, 4), "unit": "volumes/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "per_rank_ms_per_step": [round(t / args.steps * 1e3, 3) for t in per_rank],
               # host time a rank spends enqueueing one step onto an idle device (Python + ~600 C-ABI launches in the eager form, 3 launches
               # with --graph; two extra steps after the timed region); the step is GPU-bound while this stays below ms_per_step
               "per_rank_host_enqueue_ms_per_step": [round(t, 3) for t in per_rank_host],
               "communicator": {"backend": comm_backend, "world_size": comm_world, "rank_devices": rank_devices},
               "step_form": ("hip-graph replay + one flat all-reduce" + (" (uncaptured: emulator plumbing test)" if emu else "")) if graphed is not None
                            else "eager launches, bucketed all-reduce inside backward",
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[args.precision],
               "data": "synthetic" if not emu else "synthetic -- CPU-EMULATOR PLUMBING TEST, NOT A MEASUREMENT",
               "config": {"workload": f"BASELINE configs[{ {'c2': 1, 'c3': 2, 'c4': 3}[args.config] }]: {model_desc}, {'x'.join(str(v) for v in dhw)} patch, batch {B}/GPU, {args.storage + ' activation tensors' if args.storage != 'fp32' else 'fp32 tensors'}, "
                                      f"fwd + sigmoid-Dice + bwd + Adam" + (", Dropout3d on" if args.model == "unet3d" else ""),
                          "conv_arithmetic": ARITH[args.precision],
                          "activation_storage": args.storage + " (activations and their gradients between the fp32 input volume and the fp32 logits; statistics, "
                                                "weights, weight gradients and the loss fp32)" if args.storage != "fp32" else "fp32",
                          "global_batch": world * B, "parallelism": f"dp{world}"},
               "final_loss": round(loss_val, 6), "roofline": roofline, "first_layer": first_layer}
        if world == 1 and args.precision == "fp32" and not args.no_precision_modes and args.config == "c2":
            # informational: the same step with the opt-in arithmetic modes of the 3x3x3 stride-1 convs;
            # `value` above is the exact-fp32 number
            modes = {}
            for pm in ("bf16x6", "bf16x3", "bf16"):
                be.set_precision(pm)
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                dtm = (time.perf_counter() - t1) / 5
                modes[pm] = {"volumes_per_s": round(B / dtm, 3), "ms_per_step": round(dtm * 1e3, 2), "conv_arithmetic": ARITH[pm]}
                if pm == "bf16" and args.model != "dynunet":
                    # ... and with 16-bit activation storage on top (what `--precision bf16` and `--config c3` run)
                    model.act_storage = torch.bfloat16
                    for _ in range(3):
                        step()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(5):
                        step()
                    torch.cuda.synchronize()
                    dtm = (time.perf_counter() - t1) / 5
                    model.act_storage = None
                    modes["bf16 + bf16 activation storage"] = {"volumes_per_s": round(B / dtm, 3), "ms_per_step": round(dtm * 1e3, 2),
                                                               "conv_arithmetic": ARITH[pm]}
            be.set_precision("fp32")
            out["precision_modes"] = modes
        if world == 1 and args.precision == "fp32" and args.config == "c2" and args.model == "unet3d" and not args.no_c3 and not emu:
            out["c3"] = c3_block()
        if world == 1 and not args.no_cpu_baseline and args.config == "c2":
            out["cpu_baseline"] = cpu_baseline(S, args.model)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
