"""dev tool: one replayed step of a rocprofv3 kernel trace of bench.py as a per-queue timeline (start offset, duration, gap before, name).
usage: step_timeline.py TRACE.csv [which_step_from_end=3] [min_us=0]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
marks = [s for s, e, k, q in ev if "seed_advance" in k]
t0, t1 = marks[-back - 1], marks[-back]
sel = [(s, e, k, q) for s, e, k, q in ev if t0 <= s < t1]
print(f"step wall {(t1 - t0) / 1e3:.1f} us, {len(sel)} launches, summed {sum(e - s for s, e, k, q in sel) / 1e3:.1f} us")


def short(k):
    k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"^_ZN12_GLOBAL__N_1\d\d", "", k)
    return k[:48]


# all queues merged, ordered by start; column per queue keeps the picture readable
qs = sorted({q for s, e, k, q in sel})
last_end = {q: t0 for q in qs}
for s, e, k, q in sel:
    gap = (s - last_end[q]) / 1e3
    last_end[q] = max(last_end[q], e)
    if (e - s) / 1e3 < min_us and gap < 20:
        continue
    # how many OTHER kernels overlap this one's interval (concurrency seen by it)
    conc = sum(1 for s2, e2, k2, q2 in sel if q2 != q and s2 < e and e2 > s)
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  q{q:>2} gap {gap:7.1f}  ovl {conc:3d}  {short(k)}")
