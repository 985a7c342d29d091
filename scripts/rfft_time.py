"""Stand-alone timing of the framewise rFFT kernel (the kernel BASELINE's >= 40 % HBM target names) on the default
workload's geometry: 194 560 frames of 1024 samples at hop 256.  Usage: python scripts/rfft_time.py [frames] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from goofer_amd.device import Context, spec_stride

notes = 1024
n = 48510
ctx = Context(0)
ctx.plan(44100, 1024, 256)
for a in sys.argv[2:]:
    if "=" in a:
        ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
s_off = ctx.tensor(np.arange(notes + 1, dtype=np.int64) * n)
T = 1 + n // 256
f_off = ctx.tensor(np.arange(notes + 1, dtype=np.int64) * T)
F = notes * T
x = torch.randn(notes * n, device="cuda")
S = torch.empty((F, spec_stride(513)), dtype=torch.complex64, device="cuda")
for _ in range(3):
    ctx.rfft_frames(x, s_off, f_off, F, out=S)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ctx.rfft_frames(x, s_off, f_off, F, out=S)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
alg = (4 * 256 + 8 * 513) * F
print(f"{os.environ.get('GOOFER_HIP_LIB', 'default'):50s} rfft {ms:.4f} ms  {alg / ms / 1e6:.0f} GB/s alg  frac {alg / ms / 1e6 / 8000:.3f}")
