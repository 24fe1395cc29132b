#!/bin/bash
# round-2 validation run: whole GPU suite, smoke, the default bench line and the other BASELINE workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2f; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.err
for wl in kitti tiers 3dmatch-noisy 3dlomatch; do
  timeout 900 python bench.py --workload $wl --steps 32 --warmup 8 --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
python - <<PY
import json
for n in ("default","kitti","tiers","3dmatch-noisy","3dlomatch"):
    try:
        d=json.load(open("gpurun_out/r2f/bench_%s.json"%n))
        print(n, d["value"], d["ms_per_step"], d["p50_ms_per_pair_inflight1"], (d.get("p50_ms_per_pair_latency_form") or {}).get("p50_ms"), d["registered_ok"], d["work"], d["roofline"]["frac"], d["roofline_neighbour_gather"]["frac"], d["roofline_costnet"]["frac"])
        print("   ", d["stages_ms_per_pair"])
    except Exception as e:
        print(n, "FAILED", e)
PY
