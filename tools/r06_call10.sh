#!/bin/bash
# Round 6, GPU call 10: qd_temb_mlp with row groups — its tests, then CIFAR / SD / LDM A/B against the previous commit's library.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c10
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py -m gpu -x -q -k "temb or tiny or cifar_full" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
OLD=$PWD/q-diffusion_amd/lib/libqdiff_hip_prev.so
for rep in 1 2 3; do
  one "cifar previous rep=$rep" env QDIFF_HIP_LIB=$OLD timeout 600 python bench.py --model cifar --images-per-gpu 64 $X
  one "cifar row groups rep=$rep" timeout 600 python bench.py --model cifar --images-per-gpu 64 $X
done
for rep in 1 2; do
  one "sd previous rep=$rep" env QDIFF_HIP_LIB=$OLD timeout 600 python bench.py $X
  one "sd row groups rep=$rep" timeout 600 python bench.py $X
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c10/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "producers", (r.get("producer_entries") or {}).get("temb_mlp"), "box", (d.get("box") or {}).get("mfma_ubench_tops"))
PY
cat $O/ab_summary.txt
tail -3 $O/ab.err
