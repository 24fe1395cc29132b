#!/bin/bash
# latency form (keypoint tiles): parity tests, then the bench line with the latency-form measurement at 2/3 tiles
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_stages.py tests/test_gpu_pipeline.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
for t in 2 3; do
  timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --latency-tiles $t > $OUT/bench_t$t.json 2> $OUT/bench_t$t.err; tail -c 300 $OUT/bench_t$t.err
done
python - <<PY
import json
for n in ("t2","t3"):
    try:
        d=json.load(open("gpurun_out/r2g/bench_%s.json"%n))
        print(n, d["value"], d["p50_ms_per_pair_inflight1"], d["p50_ms_per_pair_latency_form"], d["registered_ok"])
        print("   ", d["stages_ms_per_pair"])
    except Exception as e:
        print(n, "FAILED", e)
PY
