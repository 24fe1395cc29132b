#!/bin/bash
# round 4: swizzled V planes + packed-f32 transform together vs the round-4 kernel before them, three alternating runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "conv or desc" 2>&1 | tail -2
V=$PWD/buffer-x_amd/csrc/variants
for i in 1 2 3; do
timeout 300 python tools/bench_conv_layers.py --K 5000 --iters 10 --tag "swizzled V + pk transform" 2>&1 | tail -1 | tee -a $OUT/vsw2.jsonl
BX_HIP_SO=$V/libbufferx_vpad.so timeout 300 python tools/bench_conv_layers.py --K 5000 --iters 10 --tag "before" 2>&1 | tail -1 | tee -a $OUT/vsw2.jsonl
done
