"""dev tool: 0.5 ms-bucket timeline per HW queue of one timed step in a rocprofv3 kernel trace of bench.py.  usage: coarse_timeline.py TRACE.csv"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows))
marks = [s for s, e, k, q in ev if "seed_advance" in k]
t0, t1 = marks[-3], marks[-2]
sel = [x for x in ev if t0 <= x[0] < t1]
def short(k):
    k = re.sub(r"^_ZN12_GLOBAL__N_1\d\d", "", k)
    return k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:46]
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, {len(sel)} launches")
for q in sorted({x[3] for x in sel}):
    print("QUEUE", q)
    b = collections.defaultdict(list)
    for x in sel:
        if x[3] == q:
            b[int((x[0] - t0) / 5e5)].append(x)
    for k in sorted(b):
        xs = b[k]
        topk = max(xs, key=lambda x: x[1] - x[0])
        print(f" {k * 0.5:5.1f} ms: n={len(xs):3d} busy={sum(e - s for s, e, _, _ in xs) / 1e3:6.0f}us top={short(topk[2])} {(topk[1] - topk[0]) / 1e3:.0f}us | first={short(xs[0][2])}")
