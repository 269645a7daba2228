#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_engine_models.py -m gpu -q -x -k "prepared_context" 2>&1 | grep -v "^Loading\|^Initializing" | grep -E "Error|assert|passed|failed" | cut -c1-600 | tail -12
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "attention or attn" 2>&1 | tail -3
