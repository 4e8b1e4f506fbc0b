"""tools/host_enqueue.py (developer tool): the host side of a training step with the emulator's launches skipped (MI355_EMU_NOEXEC=1).
Checks that the tool runs the real module / loss / optimizer path end to end on the CPU and that the no-exec switch really skips the
kernels (a 32^3 step on the emulator takes seconds when they run)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_enqueue_tool_runs_without_a_gpu():
    env = dict(os.environ)
    env.pop("MI355_EMU_NOEXEC", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_enqueue.py"), "--size", "32", "--steps", "3"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"launches skipped\): min ([0-9.]+) ms, median ([0-9.]+) ms", out.stdout)
    assert m, out.stdout
    assert float(m.group(1)) < 500.0          # ~10 ms of Python; with the kernels executed a step takes seconds here
