"""Summary of tools/gemm_pmc.sh: python tools/gemm_pmc_summary.py <dir> case ...   (per launch of gemm256_kernel, averaged over the
dispatches of each pass).  SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles (x 4 = cycles), SQ_VALU_MFMA_BUSY_CYCLES and
SQ_LDS_* in cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, cycle constants)."""
import collections, csv, glob, sys

root, cases = sys.argv[1], sys.argv[2:]
KERNEL = "gemm256"


def counters(d):
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in per.items()}     # the first two launches are warm-up


def duration(d):
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Name"]:
                return float(r["AverageNs"]) * 1e-3, r["Name"][:60]
    return float("nan"), "?"


for c in cases:
    v = {}
    for p in "abc":
        v.update(counters("%s/%s/%s" % (root, c, p)))
    us, name = duration("%s/%s/t" % (root, c))
    if not v:
        print(c, "no data"); continue
    g = v.get
    waves = g("SQ_WAVES", 2048.0)
    wc = 4 * g("SQ_WAVE_CYCLES", 0) / waves
    kcyc = g("GRBM_GUI_ACTIVE", 0) / 8
    print("== %s  %.1f us  (%s)" % (c, us, name))
    print("   waves %d | wave lifetime %.0f cycles | kernel %.0f cycles (GRBM/8) -> %.2f GHz" % (waves, wc, kcyc, kcyc / us * 1e-3 if us == us else 0))
    tot = g("SQ_WAVE_CYCLES", 1)
    print("   wave cycles: issuing %.1f %% | waiting to issue %.1f %% (of which LDS-issue %.1f %%) | at s_waitcnt/barrier %.1f %%" % (
        100 * g("SQ_ACTIVE_INST_ANY", 0) / tot, 100 * g("SQ_WAIT_INST_ANY", 0) / tot, 100 * g("SQ_WAIT_INST_LDS", 0) / tot, 100 * g("SQ_WAIT_ANY", 0) / tot))
    print("   issue cycles by type (%% of wave cycles): VALU %.1f  LDS %.1f  VMEM %.1f  SCA %.1f  MISC %.1f" % tuple(
        100 * g(k, 0) / tot for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC")))
    print("   per wave: VALU %.0f (MFMA %.0f)  LDS %.0f  VMEM %.0f  SALU %.0f | VMEM issue %.0f cycles per instruction" % (
        g("SQ_INSTS_VALU", 0) / waves, g("SQ_INSTS_MFMA", 0) / waves, g("SQ_INSTS_LDS", 0) / waves, g("SQ_INSTS_VMEM", 0) / waves,
        g("SQ_INSTS_SALU", 0) / waves, 4 * g("SQ_INST_CYCLES_VMEM", 0) / max(1.0, g("SQ_INSTS_VMEM", 1))))
    simd_cyc = 1024 * kcyc
    print("   MFMA pipe busy %.1f %% of SIMD-cycles (%.1f cycles per MFMA) | LDS array active %.1f %% of CU-cycles, bank-conflict cycles %.2f %% of active" % (
        100 * g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, simd_cyc), g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, g("SQ_INSTS_MFMA", 1)),
        100 * g("SQ_LDS_IDX_ACTIVE", 0) / max(1.0, 256 * kcyc), 100 * g("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, g("SQ_LDS_IDX_ACTIVE", 1))))
    print("   raw:", {k: round(x) for k, x in sorted(v.items())})
