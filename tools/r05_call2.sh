#!/bin/bash
# Round 5, GPU call 2: the whole GPU suite, attention rendezvous A/B (QD_ATTN_SYNC), streams + as-script with the fixes of call 1,
# K = 320 launch times old / new library, per-kernel breakdowns of the graph-replayed evaluation in both streams.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_c2
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
for rep in 1 2; do
  for sync in 1 2; do
    echo "== QD_ATTN_SYNC=$sync rep$rep" >> $O/attn_ab.log
    QD_ATTN_SYNC=$sync python tools/bench_attn.py 10 "sd self 64x64" >> $O/attn_ab.log 2>&1
    BENCH_ATTN_FLAT=1 QD_ATTN_SYNC=$sync python tools/bench_attn.py 10 "sd self 64x64" >> $O/attn_ab.log 2>&1
  done
done
cat $O/attn_ab.log
B="python bench.py --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
R04=$PWD/q-diffusion_amd/lib/libqdiff_hip_r04.so
for rep in 1 2; do
  one "fp32 rep$rep"               env $B
  one "fp32 attn-sync1 rep$rep"    env QD_ATTN_SYNC=1 $B
  one "fp16 rep$rep"               env $B --stream fp16
  one "as-script rep$rep"          env python bench.py --as-script
done
one "as-script no speculation" env QDIFF_CTX_SPECULATE=0 python bench.py --as-script
one "fp32 r04lib" env QDIFF_HIP_LIB=$R04 $B
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r05_c2/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        cl={k:v["ms"] for k,v in r.get("by_launch_class",{}).items()}
        print(name, "ms_per_step", d.get("ms_per_step"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"), cl,
              {k:d[k] for k in ("context_chain_runs_in_run","contexts_recognised_by_value","graphs_captured","wrong_speculations") if k in d})
PY
cat $O/ab_summary.txt
# K = 320 level-1 projection and friends: old / new library
for lib in "" $R04; do
  echo "== lib=${lib:-new}" >> $O/igemm_short.log
  QDIFF_HIP_LIB=$lib IGEMM_ONLY="proj|ff out|c3 320->320 @64|geglu" python tools/bench_igemm.py 4 20 >> $O/igemm_short.log 2>&1
done
echo "== new, fp16 out" >> $O/igemm_short.log
IGEMM_OUT=fp16 IGEMM_ONLY="proj|ff out|c3 320->320 @64" python tools/bench_igemm.py 4 20 >> $O/igemm_short.log 2>&1
cat $O/igemm_short.log
# per-kernel breakdown of the graph-replayed evaluation, both streams
for st in fp32 fp16; do
  QDIFF_STREAM=$st timeout 600 rocprofv3 --kernel-trace -d $O -o evb_$st -- python tools/eval_breakdown.py run sd 8 3 graph pin > $O/evb_$st.log 2>&1
  db=$(find $O -name "evb_${st}_results.db" | head -1)
  python tools/eval_breakdown.py join $db 3 > $O/sd_eval_breakdown_graph_$st.txt; head -45 $O/sd_eval_breakdown_graph_$st.txt | cut -c1-160
  [ $st = fp32 ] && python tools/eval_breakdown.py timeline $db 3 $O/sd_eval_timeline.tsv
done
find $O -name '*.db' -delete
