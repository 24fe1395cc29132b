#!/bin/bash
# round 4: shipped F(4x4) kernel with the ring slot read in place + unpredicated piece loads vs the ring-copy form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4w; mkdir -p $OUT
export BX_W43H=0
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "conv or desc" 2>&1 | tail -2
V=$PWD/buffer-x_amd/csrc/variants
for i in 1 2; do
timeout 300 python tools/bench_conv_layers.py --K 5000 --iters 10 --tag "inplace ring + unpredicated pieces" 2>&1 | tail -1 | tee -a $OUT/ring.jsonl
BX_HIP_SO=$V/libbufferx_ringcopy.so timeout 300 python tools/bench_conv_layers.py --K 5000 --iters 10 --tag "ring copy + unpredicated pieces" 2>&1 | tail -1 | tee -a $OUT/ring.jsonl
done
