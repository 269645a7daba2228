#!/bin/bash
# Round 4, GPU call 9: the opt-in fp16 activation stream against the fp32 stream at the round's final state, A/B/A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_c9; mkdir -p $out
for st in fp32 fp16 fp32 fp16; do
  echo "== SD bench --stream $st"
  timeout 400 python bench.py --stream $st --steps 20 --warmup 3 --no-cpu-baseline --no-denominators --no-extras 2> $out/bench_$st.err | tee $out/bench_$st.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['dtype'], d['roofline']['igemm_ms_per_eval'])"
done
