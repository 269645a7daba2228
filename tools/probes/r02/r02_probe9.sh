#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p9; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
IGEMM_ONLY='c3 320->320 @64|geglu|c1 320->320|ff out|lin 640|lin 1280' python tools/bench_igemm.py 4 20 2>&1 | grep -v amdgpu
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-denominators > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.json | cut -c1-330
