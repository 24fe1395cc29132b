#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kiss.py tests/test_gpu_stages.py tests/test_gpu_headline.py tests/test_gpu_pre.py -x -q -k "kiss or fps or walk or hand_computed" > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
