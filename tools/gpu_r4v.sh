#!/bin/bash
# round 4: staged grid build behind the early exit + patch_features changes: parity of everything that touches them, then bench lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4v; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pipeline.py tests/test_gpu_kiss.py tests/test_gpu_ball_epochs.py tests/test_gpu_degenerate.py tests/test_gpu_tiles.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q -k "tiers or early or headline_vs" 2>&1 | tail -3
for wl in tiers 3dmatch; do
  timeout 600 python bench.py --workload $wl --steps 48 --warmup 16 --no-cpu-baseline --e2e-pairs 0 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  python - $OUT/bench_$wl.json $wl <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "ms", d["ms_per_step"], "p50", d["p50_ms_per_pair"], "ok", d["registered_ok"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
