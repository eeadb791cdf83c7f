"""dev tool: per-queue busy time / gaps / launches of the timed steps in a rocprofv3 kernel trace of bench.py.  usage: queue_report.py TRACE.csv [nsteps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
marks = [s for s, e, k, q in ev if "seed_advance" in k][-n - 1:]
t0, t1 = marks[0], marks[-1]
sel = [(s, e, k, q) for s, e, k, q in ev if t0 <= s < t1]
print(f"wall {(t1 - t0) / 1e6 / n:.2f} ms/step, launches/step {len(sel) / n:.0f}, summed kernel time {sum(e - s for s, e, k, q in sel) / 1e6 / n:.2f} ms/step")
byq = collections.defaultdict(list)
for s, e, k, q in sel:
    byq[q].append((s, e, k))
for q, l in sorted(byq.items()):
    l.sort()
    busy = sum(e - s for s, e, k in l) / 1e6 / n
    gaps = sum(max(0, l[i + 1][0] - l[i][1]) for i in range(len(l) - 1)) / 1e6 / n
    small = sum(1 for s, e, k in l if e - s < 8000) / n
    print(f"  queue {q}: {len(l) / n:7.1f} launches/step, busy {busy:6.2f} ms, idle between its kernels {gaps:6.2f} ms, kernels < 8 us: {small:.0f}/step")
