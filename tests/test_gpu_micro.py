"""Device checks of instruction-level building blocks, as stand-alone HIP programs under scripts/micro/ (built with hipcc on the
GPU box, run as child processes)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_register_lane_transpose_on_the_device(tmp_path):
    """transpose_reg_lanehi (fft_core.h: the walkers' first FFT exchange) moves x[t] of lane (h, lo) to x[h] of lane (t, lo):
    v_permlane32_swap, v_permlane16_swap and DPP row_ror:8 under bank masks do what their comments say on this hardware."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "lane_transpose"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-Wno-unused-result", "-o", str(exe),
                    os.path.join(ROOT, "scripts", "micro", "lane_transpose.hip")], check=True, capture_output=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "transpose ok" in r.stdout, r.stdout + r.stderr
