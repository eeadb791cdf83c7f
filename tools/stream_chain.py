"""dev tool: what one stream (HSA queue) of the replayed step is made of.  usage: stream_chain.py TRACE.csv QUEUE_ID [nsteps]
Per kernel name on that queue: launches/step, summed duration, summed idle time in FRONT of those launches (gap to the previous kernel of
the same queue; under rocprofv3 the gaps are inflated by the tracer's per-dispatch host time -- read the durations, not the gaps)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
qid = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
marks = [s for s, e, k, q in ev if "seed_advance" in k][-n - 1:]
t0, t1 = marks[0], marks[-1]
sel = [(s, e, k) for s, e, k, q in ev if t0 <= s < t1 and q == qid]
agg = collections.defaultdict(lambda: [0, 0, 0])
for i, (s, e, k) in enumerate(sel):
    a = agg[k[:90]]
    a[0] += 1; a[1] += e - s
    if i:
        a[2] += max(0, s - sel[i - 1][1])
tot = sum(a[1] for a in agg.values()) / 1e6 / n
print(f"queue {qid}: {len(sel) / n:.0f} launches/step, {tot:.2f} ms of kernels/step")
for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {c / n:6.1f} x {d / c / 1e3:8.1f} us = {d / 1e6 / n:6.3f} ms   (idle in front {g / 1e6 / n:6.3f} ms)  {k}")
