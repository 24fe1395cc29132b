#!/bin/bash
# round 4: bench lines of the other BASELINE workloads (kitti = configs[2], tiers = configs[4], 3dlomatch = low-overlap half of configs[3])
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4r; rm -rf $OUT; mkdir -p $OUT
for wl in kitti tiers 3dlomatch; do
  timeout 600 python bench.py --workload $wl --steps 48 --warmup 16 --no-cpu-baseline --e2e-pairs 0 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  python - $OUT/bench_$wl.json $wl <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); ng = d["roofline_neighbour_gather"]
    print(sys.argv[2], "value", d["value"], "p50", d["p50_ms_per_pair"], "lat", d["p50_ms_per_pair_latency_form"]["p50_ms"], "ok", d["registered_ok"], "ng", ng["frac"], ng["query_kernel_frac"], "work", d["work"], "npts", d["config"]["mean_points_per_cloud"], "fps", d["fps"]["ms_per_pair"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
