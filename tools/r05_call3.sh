#!/bin/bash
# Round 5, GPU call 3: LayerNorm in the producing epilogue (qd_ln_fuse) — kernel / model tests, A/B in the whole step;
# the unmodified sampler's call pattern with both context slots warm, with and without speculative replay.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_c3
mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py tests/test_block_parity.py -q -m gpu > $O/pytest_subset.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_subset.log; tail -4 $O/pytest_subset.log
B="python bench.py --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
for rep in 1 2; do
  one "fp32 ln-fused rep$rep"      env $B
  one "fp32 ln-unfused rep$rep"    env QDIFF_LN_FUSE=0 $B
  one "fp16 ln-fused rep$rep"      env $B --stream fp16
done
one "fp16 ln-unfused" env QDIFF_LN_FUSE=0 $B --stream fp16
one "fp32 ln-unfused generic-ln-kernel" env QDIFF_LN_FUSE=0 QD_LN_ROWS8=0 $B
one "as-script (compare before replay)" env QDIFF_CTX_SPECULATE=0 python bench.py --as-script
one "as-script speculative" env QDIFF_CTX_SPECULATE=1 python bench.py --as-script
one "as-script (compare before replay) rep2" env QDIFF_CTX_SPECULATE=0 python bench.py --as-script
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r05_c3/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        cl={k:v["ms"] for k,v in r.get("by_launch_class",{}).items()}
        print(name, "ms_per_step", d.get("ms_per_step"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"), cl,
              {k:d[k] for k in ("context_chain_runs_in_run","contexts_recognised_by_value","graphs_captured","wrong_speculations","graph_replay_enqueue_ms") if k in d})
PY
cat $O/ab_summary.txt
