#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4l; rm -rf $OUT; mkdir -p $OUT
V=$PWD/buffer-x_amd/csrc/variants
for v in base ilv base ilv; do BX_HIP_SO=$V/libbufferx_$v.so timeout 200 python tools/bench_conv_layers.py --tag $v 2>&1 | tail -1 | tee -a $OUT/layers.jsonl; done
BX_HIP_SO=$V/libbufferx_ilv.so timeout 300 python -m pytest tests/test_gpu_stages.py -x -q -k "desc_conv_layer_exact or desc_net" 2>&1 | tail -2
