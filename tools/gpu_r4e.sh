#!/bin/bash
# round 4, call E: new tests -- parity sweep (direct vs default form, 4 workloads x 32 pairs), degenerate cases, 8 ranks on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4e; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_degenerate.py -x -q 2>&1 | tail -4
BX_SWEEP_REPORT=$OUT/sweep.jsonl timeout 1500 python -m pytest tests/test_gpu_sweep.py -q -s 2>&1 | grep -v "^$" | tail -12
timeout 1500 python -m pytest tests/test_gpu_bench.py -x -q -s -k world8 2>&1 | grep -v "^$" | tail -6
