#!/bin/bash
# round 4: F(4x4) input transform on packed-f32 VALU vs the scalar form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "conv or desc" 2>&1 | tail -2
V=$PWD/buffer-x_amd/csrc/variants
for i in 1 2; do
timeout 300 python tools/bench_conv_layers.py --K 5000 --iters 10 --tag "pk transform" 2>&1 | tail -1 | tee -a $OUT/pk.jsonl
BX_HIP_SO=$V/libbufferx_nopk.so timeout 300 python tools/bench_conv_layers.py --K 5000 --iters 10 --tag "scalar transform" 2>&1 | tail -1 | tee -a $OUT/pk.jsonl
done
