#!/bin/bash
# round 4, call J: valid F(4x4) CostNet layers -- rate, kernel times, parity sweep over the pose forms
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4j; rm -rf $OUT; mkdir -p $OUT
for pc in winograd winograd43; do
  timeout 400 python bench.py --pose-conv $pc --steps 24 --warmup 8 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 --inflight-sweep "" > $OUT/bench_$pc.json 2> $OUT/bench_$pc.err
  python - $OUT/bench_$pc.json $pc <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d["stages_ms_per_pair"]
    print(sys.argv[2], "value", d["value"], "pose_net", s.get("pose_net"), "desc", s.get("desc_conv"), "ok", d["registered_ok"], "costnet frac", d["roofline_costnet"]["frac"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --pose-conv winograd43 --steps 3 --warmup 1 --inflight 1 --distinct 4 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 --inflight-sweep "" > $OUT/bench_kt.log 2>&1
python - <<'PY'
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/r4j/kt/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if "wino43v" in name or "cost_l0" in name or "wino_pose" in name:
            n = name.replace("(anonymous namespace)::", "").replace("void ", "")
            print("%-60s %5d %10.1f us" % (n[:n.find("(")], calls, avg / 1e3 if avg > 1e5 else avg))
PY
BX_SWEEP_REPORT=$OUT/sweep_pose.jsonl timeout 1500 python -m pytest tests/test_gpu_sweep.py -q -s -k pose_forms 2>&1 | grep "SWEEP_REPORT\|passed\|failed" | cut -c1-800
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -size +20M -delete
