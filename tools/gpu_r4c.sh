#!/bin/bash
# round 4, call C: v3 output-store experiments (o1 = no stores, o2 = temporal stores) and slab write spread over the MFMA loop (v3l)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4c; rm -rf $OUT; mkdir -p $OUT
V=$PWD/buffer-x_amd/csrc/variants
echo "== parity v3l"
BX_HIP_SO=$V/libbufferx_v3l.so timeout 300 python -m pytest tests/test_gpu_stages.py -x -q -k "desc_conv_layer_exact or desc_net" 2>&1 | tail -2
BX_HIP_SO=$V/libbufferx_v3l.so timeout 400 python -m pytest tests/test_gpu_headline.py -x -q -k "group_walk" 2>&1 | tail -2
run() { v=$1; BX_HIP_SO=$V/libbufferx_$v.so timeout 200 python tools/bench_conv_layers.py --tag "$v" 2>&1 | tail -1 | tee -a $OUT/layers.jsonl; }
run v3
run v3l
BX_W43_STAMPS=1 run v3ls
BX_W43_STAMPS=1 run v3o1s
BX_W43_STAMPS=1 run v3o2s
