#!/bin/bash
# round 4, call K: whole GPU suite with the round-4 defaults (F(4x4) Desc layers, valid F(4x4) CostNet layers) + smoke + default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4k; rm -rf $OUT; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d["stages_ms_per_pair"]
print("value", d["value"], "p50", d["p50_ms_per_pair"], "ok", d["registered_ok"], "stages", s)
print("roofline", d["roofline"]["frac"], "costnet", d["roofline_costnet"]["frac"], d["roofline_costnet"]["avg_launch_ms"], "ng", d["roofline_neighbour_gather"]["frac"])
PY
