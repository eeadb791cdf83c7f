"""dev tool: from a rocprofv3 kernel trace of bench.py (graph replay), report per-step wall, union-busy time, summed kernel time,
and the top kernels by summed time.  usage: timeline.py TRACE.csv [nsteps_timed]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
# the timed region = last `n` occurrences of seed_advance (one per step)
marks = [s for s, e, k, q in ev if "seed_advance" in k]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
marks = marks[-n - 1:]
t0, t1 = marks[0], marks[-1]
sel = [(s, e, k, q) for s, e, k, q in ev if t0 <= s < t1]
wall = (t1 - t0) / 1e6 / n
tot = sum(e - s for s, e, k, q in sel) / 1e6 / n
busy, cur_s, cur_e = 0, None, None
for s, e, k, q in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"steps {n}: wall {wall:.2f} ms/step, union-busy {busy/1e6/n:.2f} ms/step, summed kernel time {tot:.2f} ms/step, launches/step {len(sel)/n:.0f}")
qs = collections.Counter(q for *_, q in sel)
print("queues:", dict(qs))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k, q in sel:
    k = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", k)[:80]
    agg[k][0] += e - s; agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t/1e6/n:7.3f} ms/step {c/n:7.1f}/step avg {t/c/1e3:8.1f} us  {k}")
