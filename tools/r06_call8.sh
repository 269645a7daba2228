#!/bin/bash
# Round 6, GPU call 8: two K-groups also behind split-K partials and the head-layout epilogues — kernel tests, then A/B against
# the K-groups-for-linear-rows-only build of the previous commit (QDIFF_HIP_LIB) on the three configurations.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c8
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_kernels.py -m gpu -x -q > $O/pytest_kernels.log 2>&1; echo "pytest rc=$?" >> $O/pytest_kernels.log
tail -4 $O/pytest_kernels.log
timeout 1200 python -m pytest tests/test_engine_models.py tests/test_block_parity.py -m gpu -x -q -k "tiny or cifar_full or sd_full" > $O/pytest_models.log 2>&1; echo "pytest rc=$?" >> $O/pytest_models.log
tail -4 $O/pytest_models.log
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
OLD=$PWD/q-diffusion_amd/lib/libqdiff_hip_k2lin.so
for rep in 1 2; do
  one "sd k-groups linear only rep=$rep" env QDIFF_HIP_LIB=$OLD timeout 600 python bench.py $X
  one "sd k-groups all rep=$rep" timeout 600 python bench.py $X
  one "cifar k-groups linear only rep=$rep" env QDIFF_HIP_LIB=$OLD timeout 600 python bench.py --model cifar --images-per-gpu 64 $X
  one "cifar k-groups all rep=$rep" timeout 600 python bench.py --model cifar --images-per-gpu 64 $X
  one "ldm k-groups linear only rep=$rep" env QDIFF_HIP_LIB=$OLD timeout 600 python bench.py --model ldm --images-per-gpu 64 --extra-batch 10 $X
  one "ldm k-groups all rep=$rep" timeout 600 python bench.py --model ldm --images-per-gpu 64 --extra-batch 10 $X
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c8/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"),
              "classes", {k: v.get("ms") for k, v in (r.get("by_launch_class") or {}).items()},
              "extra", (d.get("config") or {}).get("extra_batch",{}).get("ms_per_step"), "box", (d.get("box") or {}).get("mfma_ubench_tops"))
PY
cat $O/ab_summary.txt
tail -3 $O/ab.err
