"""MFMA utilisation per kernel from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_WAIT_ANY,
SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY (+ SQ_BUSY_CYCLES).   python tools/mfma_summary.py <counter_collection.csv>

util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles
an MFMA occupies its matrix pipe (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
The quotient is the fraction of the matrix-pipe cycles AT THE CLOCK THE KERNEL RAN AT that carried an MFMA.  wait / stall / issue are
the three disjoint shares of SQ_WAVE_CYCLES (waves parked at s_waitcnt or a barrier / issue-stalled / issuing)."""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel)(<[^(]*>)?", r["Kernel_Name"])
    if not m:
        continue
    k = m.group(1) + (m.group(2) or "").replace("alpro::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
rows = []
for k, c in agg.items():
    if not c.get("GRBM_GUI_ACTIVE"):
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    rows.append((cyc, k, cnt[k], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0), c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
                 c.get("SQ_ACTIVE_INST_ANY", 0) / wc))
tot = sum(r[0] for r in rows)
print("%-58s %6s %8s %9s %6s %6s %6s" % ("kernel", "calls", "share", "MFMAutil", "wait", "stall", "issue"))
for cyc, k, n, u, w, s, a in sorted(rows, reverse=True)[:24]:
    print("%-58s %6d %7.1f%% %8.1f%% %5.0f%% %5.0f%% %5.0f%%" % (k[:58], n, 100 * cyc / tot, 100 * u, 100 * w, 100 * s, 100 * a))
g = [r for r in rows if r[1].startswith("gemm_")]
if g:
    print("GEMM family (cycle-weighted): MFMA util %.1f%% of the matrix-pipe cycles at the clock the kernels ran at" % (100 * sum(r[0] * r[3] for r in g) / sum(r[0] for r in g)))
