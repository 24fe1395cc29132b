#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_headline.py tests/test_gpu_pipeline.py -x -q -k "ball or headline or pair or full_size" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 300 python tools/bench_stage.py ball --iters 30 2>&1 | grep stage
timeout 600 python bench.py --steps 24 --warmup 4 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2e/bench.json"))
print(d["value"], d["ms_per_step"], d["p50_ms_per_pair_inflight1"], d["registered_ok"])
print(d["roofline_neighbour_gather"])
PY
