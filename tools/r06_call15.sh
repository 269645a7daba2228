#!/bin/bash
# Round 6, GPU call 15: split-K target re-swept now that launches of at most one tile per CU run two K-groups per block
# (an unsplit 64-block launch already contracts two K slices at once).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c15
mkdir -p $O
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
for rep in 1 2; do
  for tg in 256 128 64 32; do
    one "cifar target=$tg rep=$rep" env QD_SPLITK_TARGET=$tg timeout 600 python bench.py --model cifar --images-per-gpu 64 $X
  done
done
for tg in 256 128 64; do
  one "sd target=$tg" env QD_SPLITK_TARGET=$tg timeout 600 python bench.py $X
  one "ldm target=$tg" env QD_SPLITK_TARGET=$tg timeout 600 python bench.py --model ldm --images-per-gpu 10 $X
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c15/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "igemm", (r.get("by_class") or {}).get("igemm",{}).get("ms"), "box", (d.get("box") or {}).get("mfma_ubench_tops"))
PY
cat $O/ab_summary.txt
tail -3 $O/ab.err
