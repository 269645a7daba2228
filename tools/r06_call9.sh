#!/bin/bash
# Round 6, GPU call 9: the two K-groups meet on their own LDS counters instead of the block's s_barrier — kernel tests, micro A/B and
# SD / LDM / CIFAR A/B against the s_barrier build (libqdiff_hip_k2bar.so, QD_K2_GROUPSYNC=0) of the same source.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c9
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q > $O/pytest_kernels.log 2>&1; echo "pytest rc=$?" >> $O/pytest_kernels.log
tail -4 $O/pytest_kernels.log
OLD=$PWD/q-diffusion_amd/lib/libqdiff_hip_k2bar.so
SH="16,1280,16,1280,3,1;16,640,32,640,3,1;16,1280,16,1280,1,1;16,2560,16,1280,3,1;16,640,32,640,1,1;16,1920,16,1280,3,1;16,1280,32,640,3,1;16,5120,16,1280,1,1"
for rep in 1 2; do
  echo "== s_barrier rep=$rep" >> $O/igemm_ab.txt; QDIFF_HIP_LIB=$OLD IGEMM_SHAPES="$SH" timeout 300 python tools/bench_igemm.py 4 20 2>/dev/null | grep custom >> $O/igemm_ab.txt
  echo "== group counters rep=$rep" >> $O/igemm_ab.txt; IGEMM_SHAPES="$SH" timeout 300 python tools/bench_igemm.py 4 20 2>/dev/null | grep custom >> $O/igemm_ab.txt
done
cat $O/igemm_ab.txt
timeout 900 python -m pytest tests/test_engine_models.py tests/test_block_parity.py -m gpu -x -q -k "tiny or cifar_full" > $O/pytest_models.log 2>&1; echo "pytest rc=$?" >> $O/pytest_models.log
tail -3 $O/pytest_models.log
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
for rep in 1 2; do
  one "sd s_barrier rep=$rep" env QDIFF_HIP_LIB=$OLD timeout 600 python bench.py $X
  one "sd group counters rep=$rep" timeout 600 python bench.py $X
  one "cifar s_barrier rep=$rep" env QDIFF_HIP_LIB=$OLD timeout 600 python bench.py --model cifar --images-per-gpu 64 $X
  one "cifar group counters rep=$rep" timeout 600 python bench.py --model cifar --images-per-gpu 64 $X
  one "ldm s_barrier rep=$rep" env QDIFF_HIP_LIB=$OLD timeout 600 python bench.py --model ldm --images-per-gpu 64 --extra-batch 10 $X
  one "ldm group counters rep=$rep" timeout 600 python bench.py --model ldm --images-per-gpu 64 --extra-batch 10 $X
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c9/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"),
              "classes", {k: v.get("ms") for k, v in (r.get("by_launch_class") or {}).items()},
              "extra", (d.get("config") or {}).get("extra_batch",{}).get("ms_per_step"), "box", (d.get("box") or {}).get("mfma_ubench_tops"))
PY
cat $O/ab_summary.txt
tail -3 $O/ab.err
