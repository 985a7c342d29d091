"""Kernel timeline of the last step of a rocprofv3 --kernel-trace run of scripts/stage_times.py.
Usage: python scripts/timeline.py <dir with *_kernel_trace.csv>   (prints start/end in ms relative to the step's first kernel)"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of k_sample_assemble starts the last step
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_sample_assemble"))
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:8.3f} {e:8.3f} {e - s:7.3f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:60]}")
