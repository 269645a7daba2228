#!/usr/bin/env python3
"""Which device kernels changed between a git revision and the working tree?

    python tools/isa_diff.py <rev> [file.hip ...]

Compiles csrc/<file> of <rev> and of the working tree for gfx950 with the flags of q-diffusion_amd/build.py
(`--save-temps`-free: `-S --cuda-device-only`), splits the device assembly into functions and compares the instruction
streams symbol by symbol (labels renumbered, comments and debug directives dropped).  Used at the end of a round whose
last commits could not be run on a GPU: a default-path kernel whose instruction stream is IDENTICAL to the last
GPU-verified revision needs no new verification; only the listed ones do.
Objects go to q-diffusion_amd/build/isa_diff/ (git-ignored)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
import build as B  # noqa: E402

OUT = os.path.join(ROOT, "q-diffusion_amd", "build", "isa_diff")


def device_asm(src_path, inc_dirs, out_path):
    cmd = [B._hipcc()] + B.CFLAGS + [f"-I{d}" for d in inc_dirs] + ["-x", "hip", "--cuda-device-only", "-S", src_path, "-o", out_path]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    return open(out_path).read()


def functions(asm):
    """{symbol: [normalised instruction lines]} of every function in a device .s file."""
    fns, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith((".L", "BB", "Lfunc")):
            name, cur = m.group(1), []
            fns[name] = cur
            continue
        if cur is None:
            continue
        s = line.split(";")[0].strip()
        if not s or s.startswith((".loc", ".file", ".cfi", ".p2align", ".Ltmp", ".Lfunc", ".size", ".type", ".section", ".text")):
            if s.startswith(".size"):
                cur = None
            continue
        cur.append(s)
    norm = {}
    for k, lines in fns.items():
        labels, out = {}, []
        for s in lines:
            m = re.match(r"^(\.LBB\w+):$", s)
            if m:
                labels.setdefault(m.group(1), f"L{len(labels)}")
        for s in lines:
            out.append(re.sub(r"\.LBB\w+", lambda mm: labels.get(mm.group(0), mm.group(0)), s))
        norm[k] = out
    return norm


def main():
    rev = sys.argv[1]
    files = sys.argv[2:] or [s for s in B.SOURCES if s.endswith(".hip")]
    os.makedirs(OUT, exist_ok=True)
    old_root = os.path.join(OUT, "old")
    os.makedirs(os.path.join(old_root, "csrc"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "include"), exist_ok=True)
    # common.h includes "../../include/qdiff_hip.h": two levels above old/csrc/ is OUT
    for rel, dst in (("q-diffusion_amd/csrc/common.h", os.path.join(old_root, "csrc", "common.h")),
                     ("include/qdiff_hip.h", os.path.join(OUT, "include", "qdiff_hip.h"))):
        open(dst, "w").write(subprocess.run(["git", "show", f"{rev}:{rel}"], cwd=ROOT, capture_output=True, text=True, check=True).stdout)
    changed_total = 0
    for f in files:
        r = subprocess.run(["git", "show", f"{rev}:q-diffusion_amd/csrc/{f}"], cwd=ROOT, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"{f}: not in {rev} (new file)")
            continue
        old_src = os.path.join(old_root, "csrc", f)
        open(old_src, "w").write(r.stdout)
        a = functions(device_asm(old_src, [os.path.join(old_root, "csrc")], os.path.join(OUT, f + ".old.s")))
        b = functions(device_asm(os.path.join(B.CSRC, f), [B.CSRC], os.path.join(OUT, f + ".new.s")))
        same = [k for k in a if k in b and a[k] == b[k]]
        diff = [k for k in a if k in b and a[k] != b[k]]
        gone, new = [k for k in a if k not in b], [k for k in b if k not in a]
        print(f"{f}: {len(same)} kernels identical, {len(diff)} changed, {len(new)} new, {len(gone)} removed")
        for k in diff:
            n = sum(1 for x, y in zip(a[k], b[k]) if x != y) + abs(len(a[k]) - len(b[k]))
            print(f"   CHANGED {k}  ({len(a[k])} -> {len(b[k])} lines, ~{n} differ)")
        for k in new:
            print(f"   new     {k}")
        for k in gone:
            print(f"   removed {k}")
        changed_total += len(diff)
    return 1 if changed_total else 0


if __name__ == "__main__":
    sys.exit(main())
