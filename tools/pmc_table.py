#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per (kernel, grid): mean counter value per dispatch.

  python tools/pmc_table.py <dir-or-csv> [substring ...]      (substrings select kernels; default igemm/attn)

Several passes (one CSV per counter set) of the same command can be given as one directory: dispatches are
matched by (kernel name, grid size, workgroup size), which is how the micro-benchmarks separate layer shapes.
Derived columns are printed when their inputs are present (quad-cycle units of the SQ counters, see
MI355X_MICROARCH.md "rocprofv3 PMC slots")."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main(path, subs):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if subs and not any(s in k for s in subs):
                continue
            key = (k, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
            a = agg[key][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for key in sorted(agg):
        d = {c: v[0] / v[1] for c, v in agg[key].items()}
        n = max(v[1] for v in agg[key].values())
        print(f"{key[0]}  grid={key[1]} wg={key[2]}  ({n} dispatches)")
        for c in sorted(d):
            print(f"    {c:34s} {d[c]:18.1f}")
        g = d.get
        if g("SQ_WAVE_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CYCLES"):
            pass
        if g("SQ_INSTS_MFMA") and g("SQ_INSTS_VALU") is not None:
            print(f"    {'-> VALU (non-MFMA) per MFMA':34s} {(d['SQ_INSTS_VALU'] - d['SQ_INSTS_MFMA']) / d['SQ_INSTS_MFMA']:18.2f}")
        if g("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
                if g(c) is not None:
                    print(f"    {'-> ' + c + ' / WAVE_CYCLES':34s} {d[c] / d['SQ_WAVE_CYCLES']:18.3f}")
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
            # MFMA_BUSY counts cycles summed over SIMDs (4 per CU, 256 CUs).  rocprofv3 (ROCm 7.2) reports GRBM_GUI_ACTIVE summed
            # over the 8 XCDs (a 0.95 ms kernel reads 16.5 M = 8 x 2.06 M cycles): kernel cycles = GUI_ACTIVE / 8
            print(f"    {'-> MFMA busy (of 1024 SIMDs)':34s} {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * d['GRBM_GUI_ACTIVE'] / 8.0):18.3f}"
                  f"   (kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs)")
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
            print(f"    {'-> L2 hit rate':34s} {d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum']):18.3f}")
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            print(f"    {'-> LDS conflict / active':34s} {d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']:18.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:] or ["igemm", "attn_kernel", "splitk"])
