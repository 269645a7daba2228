#!/usr/bin/env python3
"""VGPR / spill / scratch / SGPR / LDS of every kernel of one csrc file, as hipcc compiles it for gfx950.

    python tools/kernel_resources.py igemm_dma.hip [substring ...]

Compiles the working tree's csrc/<file> to device assembly with the flags of q-diffusion_amd/build.py and reads the
amdhsa metadata (what rocprofv3's kernel trace reports per dispatch, without a GPU).  Used to catch a template instantiation
that starts to spill after an edit of the shared body."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
import build as B  # noqa: E402


def resources(asm):
    out = {}
    for b in asm.split("  - .agpr_count:")[1:]:
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", b).group(1))
        out[re.search(r"\.name:\s+(\S+)", b).group(1)] = dict(vgpr=g("vgpr_count"), spill=g("vgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                                                              sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"))
    return out


def main():
    src = sys.argv[1]
    filt = sys.argv[2:]
    outdir = os.path.join(ROOT, "q-diffusion_amd", "build", "isa_diff")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, src + ".res.s")
    cmd = [B._hipcc()] + B.CFLAGS + [a for a in os.environ.get("QD_EXTRA_CFLAGS", "").split() if a] + \
          ["-x", "hip", "--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-3000:])
    for name, d in sorted(resources(open(out).read()).items()):
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        short = short.replace("(anonymous namespace)::", "").replace("((anonymous namespace)::ConvD)", "")
        if filt and not any(f in short for f in filt):
            continue
        print(f"{short:70s} vgpr {d['vgpr']:3d} spill {d['spill']:3d} scratch {d['scratch']:4d} B sgpr {d['sgpr']:3d} lds {d['lds']}")


if __name__ == "__main__":
    main()
