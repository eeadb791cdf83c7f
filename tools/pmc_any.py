"""dev tool: per-kernel sums of the counters of a rocprofv3 --pmc pass.  usage: pmc_any.py PMC_DIR [name filter]"""
import collections, csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if flt in row["Kernel_Name"]:
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    n = len(next(iter(cs.values())))
    print(k[:100], "launches", n)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / n:16.0f} per launch")
