#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for inf in 16 24 32 16 24 32; do
timeout 300 python bench.py --inflight $inf --steps 48 --warmup 16 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 --inflight-sweep "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', d['config']['pairs_in_flight_per_gpu'], 'value', d['value'], 'host_ms', d['host_ms_per_pair'])"
done
