"""Fold the rocprofv3 outputs of scripts/collect_profiles.sh into the small files committed under profiles/:
<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_traffic.json (per-kernel mean FETCH_SIZE / WRITE_SIZE in KB)."""
import csv, glob, json, os, shutil, sys, collections

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "profiles_" + tag)
os.makedirs(dst, exist_ok=True)
line = open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))


def short(name):
    return name.split("(")[0].strip()


res = collections.defaultdict(dict)
for which, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    vals = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(out, which, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] == key:
                vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    for k, v in vals.items():
        res[k][key + "_KB"] = sum(v) / len(v)
        res[k]["launches_sampled"] = len(v)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goofer_amd.build import source_hash
meta = {"csrc_sha256": source_hash(), "workload": bench["config"]["workload"], "frames": bench["config"]["frames_per_gpu"], "samples": bench["config"]["samples_per_gpu"],
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (only --kernel-trace beside them); per-launch means in KB",
        "correction": "gfx950: FETCH_SIZE reports half of a coalesced streaming read (MI355X_MICROARCH.md, HBM) -> bytes = (2*FETCH + WRITE)*1024"}
# HBM bytes of one step (= one goofer_render_batch).  The profiled command also times the assembly alone and the stand-alone
# rFFT, so launches are not simply "per step": a kernel launched at least once per step counts once, the once-per-process
# plan kernels pro rata, and the kernels that only the extra stages launch are left out.
# A fixed job (configs 4 / 5 with --job-notes) runs `subs` sub-batches per pass: a per-step kernel is launched `subs` times per
# pass, and the pass moves the SUM of those launches (round 3 reported the mean sub-batch as "per pass").
subs = max(1, int(bench["config"].get("sub_batches_per_gpu", 1)))
steps = max(1, res.get("k_note_finish", res.get("k_sample_assemble", {})).get("launches_sampled", 1)) / subs
stems = any(k.startswith("void k_harm_stem") for k in res)
fused_warp = any(k.startswith(("void k_env_loop<true", "void k_env_rows<true")) for k in res)
tot, per_kernel = 0.0, {}
for k, v in res.items():
    if not k.startswith(("k_", "void k_")) or "FETCH_SIZE_KB" not in v or "WRITE_SIZE_KB" not in v:
        continue
    if (stems and k.startswith("void k_rfft_frames")) or (fused_warp and k.startswith(("void k_env_loop<false", "void k_env_rows<false", "void k_row_recs<false"))):
        continue
    per_step = min(float(subs), v["launches_sampled"] / steps)            # launches of this kernel inside one pass
    per_kernel[k] = (2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0 * per_step
    tot += per_kernel[k]
meta["step_hbm_bytes"] = tot
meta["step_hbm_bytes_by_kernel"] = {k: round(v) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])}
meta["sub_batches_per_pass"] = subs
meta["step_hbm_bytes_note"] = ("sum over the kernels of one pass (= one goofer_render_batch per sub-batch, sub_batches_per_pass of them) of (2*FETCH + WRITE)*1024 per launch x launches per pass; kernels the bench launches "
                               "outside the step (stand-alone rFFT, assembly-only timing) are left out, plan-time kernels counted pro rata")
json.dump({"_meta": meta, "kernels": {k: v for k, v in sorted(res.items()) if k.startswith(("k_", "void k_"))}},
          open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
print("wrote", os.listdir(dst))
