"""Per-kernel matrix-core utilisation from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE) -> profiles/*.csv.

SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs (a 16x16x32 bf16 instruction adds 16), GRBM_GUI_ACTIVE over the
8 XCDs.  util = busy / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the share of the launch's SIMD-cycles with the matrix
pipe busy."""
import collections, csv, glob, sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    if m.get("SQ_INSTS_MFMA", 0) <= 0:
        continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    rows.append((k, n, m["SQ_INSTS_MFMA"], m["SQ_VALU_MFMA_BUSY_CYCLES"], cyc, m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)))
rows.sort(key=lambda r: -r[1] * r[4])
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches", "SQ_INSTS_MFMA_mean", "SQ_VALU_MFMA_BUSY_CYCLES_mean", "gpu_cycles_mean", "mfma_util"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.0f}", f"{r[3]:.0f}", f"{r[4]:.0f}", f"{r[5]:.3f}"])
for r in rows[:30]:
    print(f"{r[0][:90]:90s} n={r[1]:4d} util={r[5]:.3f}")
