#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p13; mkdir -p $out
timeout 900 python -m pytest tests/test_engine_models.py -m gpu -q -k "sd_tiny or sd_full or graph or plms" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
