#!/bin/bash
# round 4: patch_features_kernel variants: parity of the default build, then per-variant stage time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4u; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pipeline.py -x -q -k "patch or pipeline or register" 2>&1 | tail -3
V=$PWD/buffer-x_amd/csrc/variants
for n in ${PF_VARIANTS:-pf_base pf_r0}; do
  echo "== $n"
  BX_HIP_SO=$V/libbufferx_$n.so timeout 300 python tools/bench_stage.py patch --iters 10 2>&1 | grep '"stage"' | tee $OUT/$n.jsonl
done
echo "== shipped"
timeout 300 python tools/bench_stage.py patch --iters 10 2>&1 | grep '"stage"' | tee $OUT/shipped.jsonl
