"""GPU: the fixed-job path of bench.py end to end on ONE GPU — what the driver runs at 1, 2, 4 and 8 GPUs for BASELINE
configs 4 / 5 (`--job-notes`): LPT assignment, sub-batches ordered by length, the per-rank frame counts, the imbalance and the
optional ragged gather in the printed line.  A child process, like the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_job_mode_line_on_one_gpu():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(here, "bench.py"), "--gpus", "1", "--job-notes", "2048", "--config", "4", "--gather",
           "--no-cpu-baseline", "--no-variants", "--no-host-inclusive", "--steps", "2", "--warmup", "1", "--sub-batch", "1024"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=here,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    from goofer_amd import synthetic as syn
    frames = sum(syn.config_note_frames(4, i) for i in range(2048))
    assert line["scaling"] == "strong" and line["n_gpus"] == 1
    assert line["per_rank_frames"] == [frames] and line["imbalance"] == 1.0
    assert line["config"]["notes_per_gpu"] == 2048 and line["config"]["sub_batches_per_gpu"] == 2
    assert line["config"]["frames_per_gpu"] == frames
    g = line["gather_to_rank0"]
    assert g["bytes"] == 0 and g["ms"] >= 0.0                      # one rank: nothing crosses a link, the call still runs
    assert line["value"] > 1e6 and abs(line["value"] - frames * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-6 * line["value"]
    assert line["roofline"]["kernel"] and line["roofline"]["frac"] > 0.0
