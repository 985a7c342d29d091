import sys, time, torch
sys.path.insert(0, '/root/repo')
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload
ctx = Context(0)
wl = SamplerWorkload(ctx, 3, list(range(1024)))
for _ in range(5): wl.step()
torch.cuda.synchronize()
def run(prof, n=20):
    if prof: ctx.profile_begin(n)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): wl.step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
    if prof: ctx.profile_end()
    return dt*1e3
for rep in range(3):
    print("events on %.4f ms   events off %.4f ms" % (run(True), run(False)))
