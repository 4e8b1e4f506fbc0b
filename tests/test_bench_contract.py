"""bench.py's driver contract, checked without a GPU: flag defaults, the cpu_baseline leg, the PMC traffic look-up and the
shape of the committed bench line (profiles/r1_bench_fp32.json, produced by `python bench.py` on an MI355X)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flag_defaults(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.batch, a.size, a.precision, a.model) == (1, 2, 128, "fp32", "unet3d")      # BASELINE configs[1]
    assert a.steps >= 1 and a.warmup >= 1
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 2)


def test_cpu_baseline_leg_reports_the_contract_fields():
    r = _bench().cpu_baseline(32)          # the full 32^3 step (1 warm-up + 3 timed): a second of CPU work
    assert {"value", "unit", "cores", "kind", "sample"} <= set(r)          # the contract fields ...
    assert {"seconds_per_step", "forward_s", "backward_s", "optimizer_s", "per_iteration_s"} <= set(r)      # ... and the SURVEY 8(d) split
    assert r["kind"] == "port" and r["unit"] == "volumes/s" and r["value"] > 0 and r["cores"] >= 1
    assert "32^3" in r["sample"] and "1 warm-up + 3 timed" in r["sample"] and len(r["per_iteration_s"]) == 3
    assert r["extrapolated"] is False and "EXTRAPOLATED" not in r["sample"] and "scaled" not in r["sample"]      # the stated shape is the timed shape
    assert abs(r["forward_s"] + r["backward_s"] + r["optimizer_s"] - r["seconds_per_step"]) < 0.25 * r["seconds_per_step"]


def test_cpu_baseline_marks_an_extrapolation_and_never_scales_adam(monkeypatch):
    """Only a host that cannot hold the full patch times the half-edge patch; the line says so and the optimizer time (parameter-sized)
    is not multiplied by the voxel ratio."""
    b = _bench()
    from oracle import unet3d_ref as R
    real = R.synthetic_case

    def case(n, c, dhw, *a, **k):
        if dhw[0] == 64:
            raise MemoryError("host too small (injected)")
        return real(n, c, dhw, *a, **k)
    monkeypatch.setattr(R, "synthetic_case", case)
    r = b.cpu_baseline(64)
    assert r["extrapolated"] is True and "EXTRAPOLATED" in r["sample"] and "32^3" in r["sample"]
    # forward / backward x 8, Adam x 1: the per-iteration totals are consistent with that split
    assert abs(sum(r["per_iteration_s"]) / 3 - r["seconds_per_step"]) < 1e-2 * r["seconds_per_step"] + 2e-3
    assert r["optimizer_s"] < 0.5 * (r["forward_s"] + r["backward_s"])


def test_pmc_traffic_lookup_checks_provenance(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed rocprofv3 PMC summary -- but only from one collected on THESE kernel sources: the
    summary's first line records their sha256 (tools/pmc_summary.py) and bench.py refuses a stale file instead of quoting it.
    The family figure is call-weighted over every instantiation of the family (the verdict's one-division reproduction)."""
    b = _bench()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_summary import kernel_source_hash
    body = ('Kernel,Calls,FETCH_SIZE_MiB_per_launch,WRITE_SIZE_MiB_per_launch,fetch_x2_MiB_per_launch,avg_ms_under_pmc\n'
            '"void conv3d_wino2d_d8<0, 2>(WinoArgs)",100,348.84,183.80,697.69,0.6220\n'
            '"void conv3d_wino2d_d8<1, 1>(WinoArgs)",84,279.22,135.53,558.45,0.5320\n'
            '"void conv3d_wino2d_d8<1, 0>(WinoArgs)",16,592.89,275.63,1185.78,1.1347\n'
            '"void conv3d_wgrad_wino_ring<1>(WWRArgs)",100,149.41,27.00,298.81,0.5333\n')
    good = tmp_path / "good.csv"
    good.write_text(f"# provenance: git_sha=abc kernel_source_sha256={kernel_source_hash()}\n" + body)
    monkeypatch.setitem(b.PMC_FILES, ("c2", "fp32"), str(good))
    rows, src = b.pmc_rows("c2", "fp32")
    assert len(rows) == 4 and "FETCH_SIZE x2" in src
    traffic, hit = b.family_traffic(rows, "conv3d_wino2d")
    expect = (100 * (697.69 + 183.80) + 84 * (558.45 + 135.53) + 16 * (1185.78 + 275.63)) / 200 * 1048576
    assert len(hit) == 3 and abs(traffic - expect) < 1.0
    traffic, hit = b.family_traffic(rows, "conv3d_wgrad_wino_ring (+reduce)")
    assert len(hit) == 1 and abs(traffic - (298.81 + 27.00) * 1048576) < 1.0
    assert b.family_traffic(rows, "conv3d_k3_bf16<...>") == (None, [])
    assert b.pmc_rows("c5", "fp32") == (None, None)
    stale = tmp_path / "stale.csv"
    stale.write_text("# provenance: git_sha=abc kernel_source_sha256=0000\n" + body)
    monkeypatch.setitem(b.PMC_FILES, ("c2", "fp32"), str(stale))
    rows, src = b.pmc_rows("c2", "fp32")
    assert rows is None and "stale" in src
    old = tmp_path / "old.csv"
    old.write_text(body)                                  # a round-1 style file without provenance
    monkeypatch.setitem(b.PMC_FILES, ("c2", "fp32"), str(old))
    assert b.pmc_rows("c2", "fp32")[0] is None


@pytest.mark.parametrize("name", ["r1_bench_fp32.json", "r2_bench_fp32.json"])
def test_committed_bench_line_has_the_contract_shape(name):
    line = json.load(open(os.path.join(ROOT, "profiles", name)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["unit"] == "volumes/s" and line["dtype"] == "f32" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - line["config"]["global_batch"] * 1e3 / line["ms_per_step"]) / line["value"] < 1e-3
    roof = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["traffic"] is None or roof["traffic"] >= roof["algorithmic_bytes_per_launch"]
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0
    if name.startswith("r2"):
        # the round-2 line: traffic from a PMC summary collected on the kernel sources it was measured with, the dominant kernel's launch
        # time consistent with its share of the step, one entry per rank
        assert roof["traffic"] is not None and "stale" not in roof["traffic_source"]
        assert line["per_rank_ms_per_step"] == [line["ms_per_step"]]
        assert abs(roof["avg_launch_ms"] * roof["launches"] / 3 / line["ms_per_step"] - roof["share_of_step"]) < 0.02      # 3 roofline steps
        assert set(line["precision_modes"]) == {"bf16x6", "bf16x3", "bf16"}


@pytest.mark.parametrize("name,dtype", [("r5_bench_fp32.json", "f32"), ("r5_c3_bench.json", "bf16 (mixed)"), ("r6_bench_fp32.json", "f32"),
                                        ("r6_c3_bench.json", "bf16 (mixed)")])
def test_committed_lines_carry_the_verdict_fixes(name, dtype):
    """The round-5 lines as produced on the MI355X (tools/r5_final.sh): contract shape, a roofline whose traffic ratio a reader can redo in
    one division from the committed PMC summary, per-instantiation rows, both roofline fractions, (headline only) a cpu_baseline of the REAL
    shape and the `c3` block = BASELINE configs[2] per GPU measured warm."""
    import csv
    line = json.load(open(os.path.join(ROOT, "profiles", name)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "communicator"):
        assert k in line, k
    assert line["dtype"] == dtype and line["n_gpus"] == 1 and line["communicator"]["world_size"] == 1 and len(line["communicator"]["rank_devices"]) == 1
    assert abs(line["value"] - line["config"]["global_batch"] * 1e3 / line["ms_per_step"]) / line["value"] < 1e-3
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 2e-3
    assert roof["frac"] == (roof["mfma_pipe_frac"] if roof["bound"] == "mfma" else roof["hbm_frac"])
    assert (roof["mfma_pipe_frac"] >= roof["hbm_frac"]) == (roof["bound"] == "mfma")           # the binding roofline is the one that needs more time
    assert roof["algorithmic_bytes_per_launch"] >= roof["unfused_compulsory_bytes_per_launch"]
    assert abs(roof["traffic_over_algorithmic"] - roof["traffic"] / roof["algorithmic_bytes_per_launch"]) < 2e-3
    # the call-weighted family traffic, recomputed from the CSV the line names
    rel = roof["traffic_source"].split(" ")[0]
    rows = [r for r in csv.DictReader(open(os.path.join(ROOT, rel)).read().splitlines()[1:])]
    keys = _bench().PMC_FAMILY[roof["kernel"]]
    hit = [(int(r["Calls"]), (float(r["fetch_x2_MiB_per_launch"]) + float(r["WRITE_SIZE_MiB_per_launch"])) * 1048576) for r in rows
           if any(k in r["Kernel"] for k in keys)]
    assert abs(sum(c * b for c, b in hit) / sum(c for c, _ in hit) - roof["traffic"]) < 1.0
    inst = roof["instantiations"]
    assert len(inst) >= 3 and sum(i["launches_per_step"] for i in inst) == roof["launches_per_step"]
    assert all("traffic_over_algorithmic" in i for i in inst)
    if name == "r6_bench_fp32.json":
        # round 6: the PMC summary named by the line is the one bench.py reads for THIS tree (provenance hash of the kernel sources), the
        # roofline block carries the clock the launches ran at, the >= 256-channel launches are on conv3d_wino3d, the first layer's
        # backward is the fused kernel
        assert rel == _bench().PMC_FILES[("c2", "fp32")].replace(os.sep, "/") and _bench().pmc_rows("c2", "fp32")[0] is not None
        assert 1.5 < roof["shader_clock_ghz"] < 2.5 and roof["shader_clock_ghz"] == _bench().family_clock("c2", "fp32", roof["kernel"])
        assert "conv3d_wino3d" in roof["all_kernels"] and roof["all_kernels"]["conv3d_wino3d"]["launches"] >= 3 * 8
        fl = line["first_layer"]
        assert set(fl["kernels_ms_per_launch"]) == {"conv3d_c4_fwd", "conv3d_c4_bwd (+reduce)"} and fl["hbm_frac"] > 0.22      # round 5: 0.19
        assert line["metric"] == "training volumes/sec (128^3, 4ch->3cls)" and line["ms_per_step"] < 53.0
        # ... the 32-channel stride-2 convolution and the 1x1x1 weight gradients run their round-6 kernels, and C3 passed the verdict's 95
        assert {"conv3d_s2c32_fwd", "conv3d_s2c32_wgrad (+reduce)", "conv3d_wgrad_k1_stream (+reduce)"} <= set(roof["all_kernels"])
        assert line["c3"]["volumes_per_s_per_gpu"] > 95.0
    if name == "r6_c3_bench.json":
        # the 16-bit weight gradients: 25 launches of conv3d_wgrad_lp_tr per step (3 roofline steps), none left on the register-transposing kernel
        ak = roof["all_kernels"]
        assert ak["conv3d_wgrad_lp_tr (+reduce)"]["launches"] == 75 and not any("conv3d_wgrad_k3_bf16" in k for k in ak)
        assert "conv3d_wgrad_k1_stream (+reduce)" in ak and line["value"] > 95.0 and roof["frac"] > 0.28
    if name in ("r5_bench_fp32.json", "r6_bench_fp32.json"):
        assert "Winograd" in line["config"]["conv_arithmetic"] and roof["kernel"] == "conv3d_wino2d" and roof["bound"] == "mfma"
        assert all("conv3d_wino2d_d8<" in i["kernel"] for i in inst) and roof["frac"] > 0.62           # round 4: conv3d_wino2d_w8, 0.60
        c3 = line["c3"]
        assert set(_bench().C3_KEYS) <= set(c3) and c3["steps"] >= 20 and c3["warmup"] >= 5 and "configs[2]" in c3["workload"]
        assert abs(c3["volumes_per_s_per_gpu"] - 4e3 / c3["ms_per_step"]) / c3["volumes_per_s_per_gpu"] < 1e-3 and c3["volumes_per_s_per_gpu"] > 80.0
        cpu = line["cpu_baseline"]
        assert cpu["extrapolated"] is False and "128^3 patch" in cpu["sample"] and "scaled" not in cpu["sample"]
        assert len(cpu["per_iteration_s"]) == 3 and cpu["optimizer_s"] < 0.2 and 5.0 < cpu["seconds_per_step"] < 20.0
    else:
        assert "configs[2]" in line["config"]["workload"] and "batch 4" in line["config"]["workload"] and roof["mfma_peak_tflops"] == 2500.0
        ring = [i for i in inst if "zring" in i["kernel"]]
        assert ring and all(i["traffic_over_algorithmic"] < 1.5 for i in ring)           # the plane-ring kernels fetch every input voxel once


def test_bench_needs_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True)
    assert p.returncode != 0 and "MI355X" in p.stderr


def test_gpus_flag_launches_that_many_ranks(emu_backend):
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself and report n_gpus 2 (round 1 parsed the
    flag and ignored it). Runs the script's launch / rendezvous / reduce / timing plumbing on the CPU emulator over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "8", "--batch", "1", "--steps", "2",
                        "--warmup", "1", "--emulator-plumbing-test"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                       # rank 0 prints ONE line for the whole job
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 2
    assert len(line["per_rank_ms_per_step"]) == 2 and line["ms_per_step"] == max(line["per_rank_ms_per_step"])
    assert abs(line["value"] - 2 * 1e3 / line["ms_per_step"]) / line["value"] < 1e-2
    assert "NOT A MEASUREMENT" in line["data"]
    # proof of the ranks for the driver's SCALE line: the communicator's own world size and one device entry per rank
    com = line["communicator"]
    assert com["world_size"] == 2 and com["backend"] == "gloo" and len(com["rank_devices"]) == 2
    assert com["rank_devices"][0].startswith("rank 0") and com["rank_devices"][1].startswith("rank 1")


def test_launcher_environment_wins_over_self_launch(emu_backend):
    """Under a launcher (WORLD_SIZE set, the driver's torch.distributed.run form) bench.py must NOT spawn again: one rank given
    WORLD_SIZE=1 and --gpus 1 prints n_gpus 1."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--size", "8", "--batch", "1", "--steps", "1",
                        "--warmup", "1", "--precision", "fp16", "--emulator-plumbing-test"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["per_rank_ms_per_step"] == [line["ms_per_step"]]
    assert line["dtype"] == "f16 (mixed)" and "v_mfma_f32_32x32x16_f16" in line["config"]["conv_arithmetic"]      # the fp16 mode end to end


def test_bf16_line_says_how_activations_are_stored(emu_backend):
    """`--precision bf16` runs the mode's default 16-bit activation storage and says so in `config.activation_storage` and in the workload
    string; `--storage bf16` without bf16 operands is refused (emulator plumbing run)."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--size", "8", "--batch", "1", "--steps", "1", "--warmup", "1",
            "--emulator-plumbing-test"]
    p = subprocess.run(base + ["--precision", "bf16"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["config"]["activation_storage"].startswith("bf16") and line["dtype"] == "bf16 (mixed)", line["config"]
    assert "bf16 activation tensors" in line["config"]["workload"]
    p = subprocess.run(base + ["--storage", "bf16"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode != 0 and "--storage bf16 goes with --precision bf16" in (p.stderr + p.stdout)


@pytest.mark.parametrize("extra,form,dtype", [(["--config", "c3"], "eager launches", "bf16 (mixed)"), (["--graph"], "hip-graph replay + one flat all-reduce", "f32")])
def test_two_ranks_in_the_c3_and_the_graphed_form(emu_backend, extra, form, dtype):
    """8-rank readiness without hardware (round 5): the two other forms the driver may launch -- `--gpus N --config c3` (BASELINE configs[2]:
    bf16 operands, batch 4 per GPU, 16-bit activation storage, bucketed exchange inside backward) and `--gpus N --graph` (one exchange of
    the flat gradient buffer per step) -- through the script's own launch / rendezvous / reduce / timing plumbing at world size 2 on the
    CPU emulator over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "8", "--batch", "1", "--steps", "2",
                        "--warmup", "1", "--emulator-plumbing-test"] + extra, capture_output=True, text=True, env=env, timeout=1800)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    per_gpu = 4 if "c3" in extra else 1                 # --config c3 fixes the batch at BASELINE configs[2]'s 4 per GPU
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 2 * per_gpu
    assert line["step_form"].startswith(form) and line["dtype"] == dtype
    assert line["communicator"]["world_size"] == 2 and len(line["per_rank_ms_per_step"]) == 2
    assert abs(line["value"] - 2 * per_gpu * 1e3 / line["ms_per_step"]) / line["value"] < 1e-2
    if "c3" in extra:
        assert "configs[2]" in line["config"]["workload"] and line["config"]["activation_storage"].startswith("bf16")
    assert line["final_loss"] == line["final_loss"] and 0.0 < line["final_loss"] < 2.0      # a finite Dice loss on both forms


def test_c3_block_keeps_the_figures_of_the_child_line(monkeypatch):
    """The default N = 1 line carries BASELINE configs[2]'s per-GPU shape under `c3` (verdict round 4, item 5): a child `bench.py --config c3`
    whose line is reduced to the keys below; a failing child yields an `error` entry instead of taking the headline line down."""
    b = _bench()
    child = {"value": 84.0, "ms_per_step": 47.6, "steps": 5, "warmup": 2, "dtype": "bf16 (mixed)", "per_rank_host_enqueue_ms_per_step": [9.0],
             "config": {"workload": "BASELINE configs[2]: ...", "activation_storage": "bf16 (...)"},
             "roofline": {"bound": "mfma", "frac": 0.3, "kernel": "conv3d_k3_bf16<...>", "mfma_pipe_frac": 0.3, "hbm_frac": 0.12,
                          "traffic_over_algorithmic": 1.4, "share_of_step": 0.36}}
    seen = {}

    class R:
        returncode, stderr = 0, ""
        stdout = "noise\n" + json.dumps(child) + "\n"

    def run(cmd, **kw):
        seen["cmd"] = cmd
        return R
    monkeypatch.setattr(subprocess, "run", run)
    blk = b.c3_block()
    assert set(blk) == set(b.C3_KEYS) and blk["volumes_per_s_per_gpu"] == 84.0 and blk["mfma_pipe_frac"] == 0.3 and blk["hbm_frac"] == 0.12
    assert seen["cmd"][2:5] == ["--config", "c3", "--steps"] and "--no-cpu-baseline" in seen["cmd"]
    R.returncode = 3
    blk = b.c3_block()
    assert "error" in blk and "exited 3" in blk["error"]
