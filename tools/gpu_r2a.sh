#!/bin/bash
# round 2, call A: new headline-size parity tests + the multi-rank bench test + one bench line
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_headline.py tests/test_gpu_bench.py -q --durations=8 > gpurun_out/r2a/pytest_new.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest_new.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_headline.py --deselect tests/test_gpu_bench.py > gpurun_out/r2a/pytest_old.log 2>&1; tail -3 gpurun_out/r2a/pytest_old.log
tail -25 gpurun_out/r2a/pytest_new.log
