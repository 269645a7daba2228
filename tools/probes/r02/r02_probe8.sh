#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p8; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py -m gpu -q -x -k "temb or tiny or graph or plms or flip or cifar_full" > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $out/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-denominators > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.json | cut -c1-330; tail -3 $out/bench.err
