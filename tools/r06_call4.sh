#!/bin/bash
# Round 6, GPU call 4: grouped q / k / v launches (qd_conv2d_i8_group) — kernel tests of the whole library after the body /
# wrapper refactor, then A/B on the SD step: grouped vs single launches (QD_QKV_GROUP=0), 128-row members (QD_GROUP_MT=1), and the
# hi + lo P.V kernel compiled for four waves per SIMD (libqdiff_hip_pvf4.so) on peaked rows.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c4
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_kernels.py -m gpu -x -q > $O/pytest_kernels.log 2>&1; echo "pytest rc=$?" >> $O/pytest_kernels.log
tail -4 $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_engine_models.py tests/test_block_parity.py -m gpu -x -q -k "tiny or hooks_registered or repreparing or cifar_full" > $O/pytest_models.log 2>&1; echo "pytest rc=$?" >> $O/pytest_models.log
tail -4 $O/pytest_models.log
L=$PWD/q-diffusion_amd/lib
for rep in 1 2; do
  echo "== pvfull occ3 peaked rep=$rep" >> $O/attn_ab.txt; BENCH_ATTN_FLAT=0 timeout 300 python tools/bench_attn.py 10 "sd self 64x64" >> $O/attn_ab.txt 2>> $O/attn_ab.err
  echo "== pvfull occ4 peaked rep=$rep" >> $O/attn_ab.txt; QDIFF_HIP_LIB=$L/libqdiff_hip_pvf4.so BENCH_ATTN_FLAT=0 timeout 300 python tools/bench_attn.py 10 "sd self 64x64" >> $O/attn_ab.txt 2>> $O/attn_ab.err
done
cat $O/attn_ab.txt
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/sd_ab.log; ( "$@" ) >> $O/sd_ab.log 2>> $O/sd_ab.err; }
for rep in 1 2; do
  one "single launches rep=$rep" env QD_QKV_GROUP=0 timeout 600 python bench.py $X
  one "grouped rep=$rep" timeout 600 python bench.py $X
  one "grouped, 128-row members rep=$rep" env QD_GROUP_MT=1 timeout 600 python bench.py $X
  one "grouped + pvf4 rep=$rep" env QDIFF_HIP_LIB=$L/libqdiff_hip_pvf4.so timeout 600 python bench.py $X
done
python - <<'PY' > $O/sd_ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c4/sd_ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        cl=r.get("by_launch_class",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"), "launches", r.get("library_launches_per_eval"),
              "heads", cl.get("heads_i8_out",{}).get("ms"), "short_k", cl.get("short_k_f32_out",{}).get("ms"),
              "by_class", {k: v.get("ms") for k, v in (r.get("by_class") or {}).items()}, "replay", r.get("graph_replay_eval_ms"),
              "attn4096", (r.get("attention_calls") or {}).get("T4096_S4096_d40",{}).get("ms"), "box", (d.get("box") or {}).get("mfma_ubench_tops"), (d.get("box") or {}).get("exp_ginst_s"))
PY
cat $O/sd_ab_summary.txt
tail -5 $O/sd_ab.err
