#!/bin/bash
# round 4, last call: slab writes behind the MFMA loop in both F(4x4) kernels: parity (stages + whole pairs + real-size vs reference), layer times, then the profile set
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_headline.py -x -q -k "vs_reference" 2>&1 | tail -2
timeout 200 python tools/bench_conv_layers.py --K 5000 --iters 10 --tag "slab writes behind the loop" 2>&1 | tail -1 | tee -a $OUT/lastw.jsonl
bash tools/profile_r4.sh r04 > $OUT/profile.log 2>&1
tail -5 $OUT/profile.log
