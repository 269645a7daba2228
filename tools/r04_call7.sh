#!/bin/bash
# Round 4, GPU call 7: k / v projections of the small self-attentions on side streams next to the q projection (QDIFF_QKV_FORK), A/B/A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_c7; mkdir -p $out
for f in 0 1 0 1; do
  echo "== SD bench QDIFF_QKV_FORK=$f"
  QDIFF_QKV_FORK=$f timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-denominators --no-extras 2> $out/bench_fork$f.err | tee $out/bench_fork$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['igemm_ms_per_eval'])"
done
