"""dev tool: aggregate rocprofv3 counter_collection CSVs per kernel (mean per dispatch).  usage: pmc_agg.py DIR [name-filter]"""
import csv, glob, sys, collections, re
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if flt and flt not in k: continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/p0_kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if flt and flt not in k: continue
        dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    short = re.sub(r"\(anonymous namespace\)::", "", k)[:110]
    print(short, f"  n={len(next(iter(cs.values())))}", f" dur_us(mean/min)={sum(dur[k])/max(len(dur[k]),1):.1f}/{min(dur[k]) if dur[k] else 0:.1f}")
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c in sorted(m): print(f"    {c:28s} {m[c]:16.0f}")
    if "SQ_INSTS_MFMA" in m and m["SQ_INSTS_MFMA"] > 0:
        mf = m["SQ_INSTS_MFMA"]
        print("    per MFMA: VALU %.2f  SALU %.2f  LDS %.2f  VMEM_RD %.3f VMEM_WR %.3f | wave_cycles/MFMA %.1f  wait_any %.1f wait_inst %.1f active %.1f" % (
            m.get("SQ_INSTS_VALU", 0) / mf, m.get("SQ_INSTS_SALU", 0) / mf, m.get("SQ_INSTS_LDS", 0) / mf, m.get("SQ_INSTS_VMEM_RD", 0) / mf, m.get("SQ_INSTS_VMEM_WR", 0) / mf,
            m.get("SQ_WAVE_CYCLES", 0) / mf, m.get("SQ_WAIT_ANY", 0) / mf, m.get("SQ_WAIT_INST_ANY", 0) / mf, m.get("SQ_ACTIVE_INST_ANY", 0) / mf))
