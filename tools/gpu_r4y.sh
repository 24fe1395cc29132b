#!/bin/bash
# round 4: the real-size low-overlap fixture (headline_lo) through the headline chain
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4y; rm -rf $OUT; mkdir -p $OUT
BX_REALSIZE_REPORT=$OUT/realsize_report.jsonl timeout 1200 python -m pytest tests/test_gpu_headline.py -q -s -k "headline_lo" 2>&1 | grep -E "REALSIZE_REPORT|passed|failed|Error|assert" | cut -c1-1200
