#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2c; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2c/bench.json"))
print(d["value"], d["ms_per_step"], d["p50_ms_per_pair_inflight1"], d["registered_ok"])
print(d["roofline_neighbour_gather"])
print(d["stages_ms_per_pair"])
PY
