#!/bin/bash
# round 4: early slab requests in both F(4x4) kernels: parity of stages + whole pairs, then the bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4w; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pipeline.py tests/test_gpu_degenerate.py -x -q 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_headline.py -x -q -k "vs_reference or pose_net or desc" 2>&1 | tail -2
timeout 600 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --e2e-pairs 0 --latency-tiles 0 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "desc", d["stages_ms_per_pair"]["desc_conv"], "pose", d["stages_ms_per_pair"]["pose_net"], "frac", d["roofline"]["frac"], d["roofline_costnet"]["frac"])
PY
