#!/bin/bash
# Round 5, GPU call 9: split-K target sweep on the three configurations (one box).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_c9
mkdir -p $O
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
for t in 512 256 128 192 384 64 256 512; do one "cifar splitk-target $t" env QD_SPLITK_TARGET=$t python bench.py --model cifar --images-per-gpu 64 $X; done
for t in 512 256 128 256 512 192; do one "sd splitk-target $t" env QD_SPLITK_TARGET=$t python bench.py $X; done
for t in 512 256 128; do one "ldm splitk-target $t" env QD_SPLITK_TARGET=$t python bench.py --model ldm --images-per-gpu 64 $X; done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r05_c9/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"))
PY
cat $O/ab_summary.txt
