#!/bin/bash
# round 4, call G: neighbour gather with index epochs -- parity (epoch tests, ball stage tests, pipeline + headline chains), stage times on the three workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4g; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ball_epochs.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pipeline.py -x -q -k "ball or pair_matches or larger or kitti_scale or full_size or tiny" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_headline.py -x -q -k "descriptor_chain or vs_reference or runs_agree" 2>&1 | tail -3
for wl in 3dmatch kitti tiers; do
  timeout 400 python bench.py --workload $wl --steps 16 --warmup 4 --distinct 8 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 --inflight-sweep "" > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  python - $OUT/bench_$wl.json $wl <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); ng = d["roofline_neighbour_gather"]
    print(sys.argv[2], "value", d["value"], "ng frac", ng["frac"], "query frac", ng["query_kernel_frac"], "query ms", ng["query_kernel_avg_ms"], "build/pair", ng["grid_build_ms_per_pair"], "ok", d["registered_ok"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
