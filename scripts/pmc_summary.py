"""Average every collected counter per kernel name from a rocprofv3 --pmc CSV directory."""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in files:
    with open(fn) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            short = name.split("(")[0].split("<")[0].replace("void ", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from goofer_amd.build import source_hash
print("# csrc_sha256=" + source_hash())
for k in sorted(acc):
    if not k.startswith("k_"):
        continue
    parts = ["%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())]
    print(k, " ".join(parts), "n=%d" % len(next(iter(acc[k].values()))))
