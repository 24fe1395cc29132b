#!/bin/bash
# round 4, call H: epoch tests + phase stamps of the query kernel with / without index epochs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4h; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ball_epochs.py -x -q 2>&1 | tail -4
for wl in kitti 3dmatch; do for st in 0 1; do
  timeout 200 python tools/ball_probe.py $wl $st 2>&1 | tail -1 | tee -a $OUT/probe.jsonl
  BX_BALL_EPOCHS=1 timeout 200 python tools/ball_probe.py $wl $st 2>&1 | tail -1 | tee -a $OUT/probe.jsonl
done; done
