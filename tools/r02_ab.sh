#!/bin/bash
# usage: tools/r02_ab.sh "<ENV1>" "<ENV2>" ...   each argument is an env assignment string (may be empty) for one bench run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab; mkdir -p $out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-denominators > $out/run$i.json 2> $out/run$i.err
  echo "[$e] $(python - <<PY
import json
try:
    d=json.loads(open('$out/run$i.json').read().strip().splitlines()[-1])
    print('ms_per_step', d['ms_per_step'], 'igemm_ms', d['roofline'].get('igemm_ms_per_eval'), 'frac', d['roofline']['frac'])
except Exception as ex:
    print('FAILED', ex)
PY
)"
done
