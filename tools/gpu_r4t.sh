#!/bin/bash
# round 4, call T: the six real-size reference-minted fixtures (three new) through the capture chain
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4t; rm -rf $OUT; mkdir -p $OUT
BX_REALSIZE_REPORT=$OUT/realsize_report.jsonl timeout 2400 python -m pytest tests/test_gpu_headline.py -q -s -k "headline" 2>&1 | grep "REALSIZE_REPORT\|passed\|failed\|Error\|assert" | cut -c1-900
