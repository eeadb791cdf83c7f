"""Aggregate a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE into
per-kernel MFMA-pipe utilisation.  usage: pmc_mfma_agg.py PMC_DIR OUT.txt [min_total_cycles]

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (elapsed cycles x 1024 SIMDs), elapsed cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the sum
over the 8 XCDs; MI355X_MICROARCH.md, DVFS note).  SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per v_mfma_f32_32x32x16_bf16 and 64 per
v_mfma_f32_32x32x2_f32 on the issuing SIMD."""
import collections, csv, glob, sys
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
rows = []
for k, cs in acc.items():
    m = {c: sum(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui <= 0:
        continue
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    rows.append((gui, k, n, busy / (gui * 1024.0), m.get("SQ_INSTS_MFMA", 0.0) / n, m.get("SQ_BUSY_CU_CYCLES", 0.0) / (gui * 256.0 * 4) if gui else 0, gui / n))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
with open(out, "w") as fh:
    fh.write(f"# per-kernel MFMA pipe utilisation (eager launches, PMC pass); share = kernel's share of all elapsed GPU cycles in the pass\n")
    fh.write(f"# {'share':>6} {'launches':>8} {'cycles/launch':>13} {'MFMA insts/launch':>17} {'MfmaUtil':>8}  kernel\n")
    for gui, k, n, util, insts, cu, cyc in rows[:60]:
        fh.write(f"  {gui / tot:6.3f} {n:8d} {cyc:13.0f} {insts:17.0f} {util:8.3f}  {k[:120]}\n")
print(open(out).read()[:3000])
