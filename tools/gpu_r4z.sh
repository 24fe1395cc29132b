#!/bin/bash
# round 4: persistent grid of the convolution kernels below the CU count (FPS workgroups of other pairs hold 96 KB of LDS on up to 40 CUs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4z; rm -rf $OUT; mkdir -p $OUT
for capv in 0 240 224 208; do
  BX_CONV_PERSIST_CAP=$capv timeout 600 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --e2e-pairs 0 --latency-tiles 0 > $OUT/bench_$capv.json 2> $OUT/bench_$capv.err
  python - $OUT/bench_$capv.json $capv <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("cap", sys.argv[2], "value", d["value"], "ms", d["ms_per_step"], "p50", d["p50_ms_per_pair"])
except Exception as e:
    print("cap", sys.argv[2], "FAILED", e)
PY
done
