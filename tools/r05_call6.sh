#!/bin/bash
# Round 5, GPU call 6: per-evaluation dispatch counts of the CIFAR / LDM lines (graph-replayed evaluation), split-K target sweep.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_c6
mkdir -p $O
for m in cifar ldm; do
  timeout 400 rocprofv3 --kernel-trace -d $O -o evb_$m -- python tools/eval_breakdown.py run $m 64 3 graph > $O/evb_$m.log 2>&1
  db=$(find $O -name "evb_${m}_results.db" | head -1)
  python tools/eval_breakdown.py join $db 3 > $O/${m}_eval_breakdown_graph.txt; head -30 $O/${m}_eval_breakdown_graph.txt | cut -c1-150
done
find $O -name '*.db' -delete
B="python bench.py --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
for t in 512 256 768 1024 512; do
  one "fp32 splitk-target $t" env QD_SPLITK_TARGET=$t $B
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r05_c6/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        cl={k:v["ms"] for k,v in r.get("by_launch_class",{}).items()}
        print(name, "ms_per_step", d.get("ms_per_step"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"), cl)
PY
cat $O/ab_summary.txt
