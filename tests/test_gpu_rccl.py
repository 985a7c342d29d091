"""GPU: the RCCL calls of the multi-GPU bench on ONE GPU (world size 1), so that the driver's 8-GPU run is not the first time
that code touches a device: process-group init on cuda:0, barrier, the MAX / SUM all-reduces of shard.reduce_timing on device
tensors, the all-gather of the per-rank frame counts (bench.py), the size all-gather of shard.gather_audio.  A child process:
the process group lives and dies with it."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys
sys.path.insert(0, os.environ["GOOFER_REPO"])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from goofer_amd import shard
torch.cuda.synchronize()
dist.barrier()
t, f = shard.reduce_timing(1.25, 194560, device="cuda", always=True)       # bench.py: elapsed MAX, frames SUM
assert (t, f) == (1.25, 194560), (t, f)
x = torch.tensor([194560.0], dtype=torch.float64, device="cuda")           # bench.py: per-rank frames
allf = [torch.zeros_like(x)]
dist.all_gather(allf, x)
assert [int(v.item()) for v in allf] == [194560]
mix = torch.arange(1000, dtype=torch.float32, device="cuda")
got = shard.gather_audio(mix, [400, 600], dst=0, always=True)              # sizes all-gathered on the device
assert len(got) == 1 and torch.equal(got[0][0], mix) and got[0][1] == [400, 600]
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl world-1 ok", torch.cuda.get_device_name(0))
'''


def test_rccl_calls_of_the_bench_at_world_size_one():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GOOFER_REPO=here, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=300, env=env, cwd=here)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "rccl world-1 ok" in res.stdout
