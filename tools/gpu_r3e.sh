#!/bin/bash
# round 3, call E: threaded harness (e2e rate in bench.py), radius threshold on the fly
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3e; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_harness.py tests/test_gpu_dist.py tests/test_gpu_bench.py -x -q 2>&1 | tail -6
timeout 300 python -m pytest tests/test_gpu_stages.py -x -q -k radius 2>&1 | tail -3
timeout 900 python bench.py --steps 24 --warmup 8 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3e/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "host_ms", d["host_ms_per_pair"], "stages", d["stages_ms_per_pair"])
print("e2e", json.dumps(d.get("e2e_pairs_per_s"), indent=1))
PY
tail -5 $OUT/bench.err
