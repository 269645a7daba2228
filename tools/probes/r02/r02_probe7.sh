#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p7; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -s --deselect "tests/test_block_parity.py::test_blocks_teacher_forced[sd_full]" > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 $out/pytest.log; grep -E "fastdiv|^\[(ldm|cifar)_full\] (ldm|cifar|code)" $out/pytest.log | cut -c1-220
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-denominators > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.json | cut -c1-330
