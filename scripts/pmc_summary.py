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
for k in sorted(acc):
    if not k.startswith("k_"):
        continue
    parts = ["%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())]
    print(k, " ".join(parts), "n=%d" % len(next(iter(acc[k].values()))))
