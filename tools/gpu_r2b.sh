#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2b; mkdir -p $OUT
VARIANTS=1 bash tools/gpu_conv_pmc3.sh 2>&1 | grep -E "conv32_kernel<8, 128|conv32_kernel<8, 64"
for v in 1 0; do
echo "== bench BX_CONV32=$v"
BX_CONV32=$v timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2b/bench_$v.json"))
print(d["value"], d["ms_per_step"], d["p50_ms_per_pair_inflight1"], d["roofline"]["frac"], d["stages_ms_per_pair"]["desc_conv"], d["registered_ok"])
PY
done
