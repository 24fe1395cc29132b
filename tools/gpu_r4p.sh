#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for inf in 8 16 8 16; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --inflight $inf --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 --inflight-sweep "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style inflight', d['config']['pairs_in_flight_per_gpu'], 'value', d['value'])"
done
