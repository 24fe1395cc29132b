#!/bin/bash
# round 4, call D: consolidated F(4x4) kernel for all eight layers (CW = 32 instantiations), arithmetic forms in bx_params: whole GPU suite,
# layer times, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4d; rm -rf $OUT; mkdir -p $OUT
timeout 200 python tools/bench_conv_layers.py --tag shipped 2>&1 | tail -1 | tee -a $OUT/layers.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 400 python bench.py --steps 24 --warmup 8 --e2e-pairs 0 > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d["stages_ms_per_pair"]
print("value", d["value"], "p50", d["p50_ms_per_pair"], "sweep", d["inflight_sweep"], "ok", d["registered_ok"], "npts", d["config"]["mean_points_per_cloud"])
print("stages", s)
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "algorithmic_rate_x_peak", "avg_launch_ms")})
print("costnet", {k: d["roofline_costnet"][k] for k in ("achieved", "frac", "avg_launch_ms")})
print("ng", {k: d["roofline_neighbour_gather"][k] for k in ("frac", "query_kernel_frac", "avg_launch_ms")})
PY
