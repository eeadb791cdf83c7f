"""dev tool: per-kernel launches / time per step over the timed steps of a rocprofv3 kernel trace of bench.py.  usage: kernel_hist.py TRACE.csv [nsteps] [top]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
qsel = sys.argv[4] if len(sys.argv) > 4 else None
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if qsel is None or r.get("Queue_Id") == qsel or "seed_advance" in r["Kernel_Name"]))
marks = [s for s, e, k in ev if "seed_advance" in k][-n - 1:]
t0, t1 = marks[0], marks[-1]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in ev:
    if t0 <= s < t1:
        k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"^_ZN12_GLOBAL__N_1\d\d", "", k)
        a = agg[k[:70]]
        a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values()); cnt = sum(a[0] for a in agg.values())
print(f"{cnt / n:.0f} launches/step, {tot / 1e6 / n:.2f} ms summed kernel time/step")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{a[0] / n:7.1f} x {a[1] / a[0] / 1e3:8.1f} us = {a[1] / 1e6 / n:6.3f} ms  {k}")
print("--- by launch count")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{a[0] / n:7.1f} x {a[1] / a[0] / 1e3:8.1f} us = {a[1] / 1e6 / n:6.3f} ms  {k}")
