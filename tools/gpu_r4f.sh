#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4f; rm -rf $OUT; mkdir -p $OUT
BX_SWEEP_REPORT=$OUT/sweep.jsonl timeout 1500 python -m pytest tests/test_gpu_sweep.py -q -s 2>&1 | grep -v "^$" | grep "SWEEP_REPORT\|passed\|failed\|Error" | cut -c1-900
