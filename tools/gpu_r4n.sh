#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pipeline.py tests/test_gpu_headline.py -x -q -k "radius or pair_matches or cloud_above or runs_agree or config0" 2>&1 | tail -2
timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 --inflight-sweep "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'radius', d['stages_ms_per_pair']['radius'])"
