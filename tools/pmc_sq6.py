"""Summarise the SQ counter passes of tools/pmc_sq6.sh: per workload and kernel the average per launch of every counter, and
  mfma_busy (chip)       = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 32 CUs x GRBM_GUI_ACTIVE): share of ALL the chip's SIMD cycles with an
                           MFMA in the pipe while the kernel runs (the gfx94x MfmaUtil formula; GRBM_GUI_ACTIVE comes back summed over
                           the 8 XCDs -- 113 us x 2.4 GHz x 8 for the headline's recurrent kernel -- so the CU count beside it is an XCD's);
  mfma_busy (active CUs) = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): the same over the CUs that hold a wave of the kernel
                           (a recurrent launch fills 52 - 128 of the 256 CUs by design);
  MFMA_BUSY / SQ_BUSY    = the raw ratio VERDICT r5 names (SQ_BUSY_CYCLES counts per shader engine, the ratio is not a fraction).
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16 tile); SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units")."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
CUS = 256
KEEP = ("lstm_", "gemm_")
print("# SQ counters per kernel (rocprofv3 --kernel-trace --pmc, bench.py --steps 3 --warmup 1, bf16; tools/pmc_sq6.sh)\n")
print(__doc__.split("counter, and\n")[1] + "\n")
for wl in ("timit_3x250_blstm_H125", "lvcsr_4x512_blstm_8000"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, wl + "_g*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("cn::", "")
            if any(s in k for s in KEEP):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("## %s\n" % wl)
    if not acc:
        print("(no counter files found)\n")
        continue
    cols = ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_LDS",
            "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
    print("| kernel | launches | " + " | ".join(cols) + " | mfma_busy (chip) | MFMA_BUSY / SQ_BUSY | mfma_busy (active CUs) |")
    print("|---|---|" + "---|" * (len(cols) + 3))
    for k in sorted(acc, key=lambda k: -sum(acc[k].get("GRBM_GUI_ACTIVE", [0]))):
        a = {c: (sum(v) / len(v)) for c, v in acc[k].items()}
        n = max(len(v) for v in acc[k].values())
        gui, mf, sqb, bcu = a.get("GRBM_GUI_ACTIVE"), a.get("SQ_VALU_MFMA_BUSY_CYCLES"), a.get("SQ_BUSY_CYCLES"), a.get("SQ_BUSY_CU_CYCLES")
        chip = "%.4f" % (mf / (4.0 * (CUS / 8) * gui)) if gui and mf is not None else "-"
        r2 = "%.4f" % (mf / sqb) if sqb and mf is not None else "-"
        r3 = "%.4f" % (mf / (4.0 * bcu)) if bcu and mf is not None else "-"
        print("| `%s` | %d | " % (k[:70], n) + " | ".join("%.0f" % a[c] if c in a else "-" for c in cols) + " | %s | %s | %s |" % (chip, r2, r3))
    print()
