#!/usr/bin/env python3
"""Static census of the K-step loop of the contraction kernels: instruction classes per iteration, from the gfx950 assembly
(`hipcc -save-temps`).  Usage:  python tools/isa_kstep_census.py <igemm_dma-hip-amdgcn-amd-amdhsa-gfx950.s> [filter]

The K-step is the innermost loop that contains the kernel's MFMAs: the region from the target label of the last backward
branch that encloses all `v_mfma` instructions to that branch.  Classes: MFMA, VALU (other v_*), SALU (s_* except waits /
barriers / branches), LDS (ds_*), VMEM (global_/buffer_/scratch_), WAIT (s_waitcnt, s_nop), BARRIER, BRANCH."""
import collections
import re
import sys


def klass(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "VMEM"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "WAIT"
    if op.startswith("s_barrier"):
        return "BARRIER"
    if op.startswith(("s_cbranch", "s_branch")):
        return "BRANCH"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def main():
    src = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else "igemm_kernel"
    print("| kernel (template arguments) | MFMA | VALU | SALU | LDS | VMEM | WAIT | other issues per MFMA | scratch in loop |")
    print("|---|---|---|---|---|---|---|---|---|")
    for m in re.finditer(r"^(_Z\S*%s\S*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(flt), src, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = [l.strip() for l in body.split("\n")]
        labels = {re.match(r"^(\.LBB\w+):", l).group(1): i for i, l in enumerate(lines) if re.match(r"^\.LBB\w+:", l)}
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        if not mf:
            continue
        best = None
        for i, l in enumerate(lines):
            mm = re.match(r"^s_cbranch_\S+\s+(\.LBB\S+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:          # backward branch
                lo, hi = labels[mm.group(1)], i
                inside = [k for k in mf if lo <= k <= hi]
                # the loop with the most MFMAs wins (a peeled first iteration leaves a copy outside), then the tightest one
                if inside and (best is None or (len(inside), -(hi - lo)) > (best[2], -(best[1] - best[0]))):
                    best = (lo, hi, len(inside))
        if best is None:
            continue
        ops = [l.split()[0] for l in lines[best[0]:best[1] + 1] if l and not l.startswith((".", ";")) and not l.endswith(":")]
        c = collections.Counter(klass(o) for o in ops)
        scr = sum(1 for l in lines[best[0]:best[1] + 1] if l.startswith("scratch_"))
        targs = re.search(r"igemm_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d)ELi(\d)E", name)
        tag = "<MT=%s,NT=%s,WM=%s,WN=%s,SPLIT=%s,OUT=%s,WB=%s>" % targs.groups() if targs else name[18:60]
        others = c["VALU"] + c["SALU"] + c["LDS"] + c["VMEM"] + c["WAIT"] + c["BARRIER"] + c["BRANCH"]
        print(f"| {tag} | {c['MFMA']} | {c['VALU']} | {c['SALU']} | {c['LDS']} | {c['VMEM']} | {c['WAIT']} | {others / c['MFMA']:.1f} | {scr} |")


if __name__ == "__main__":
    main()
