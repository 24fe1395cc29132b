#!/bin/bash
# round 4, call I: CostNet layers 1..5 as valid F(4x4, 3x3) convolutions (k_wino43v.hip): parity + times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4i; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_stages.py -x -q -k "pose_conv_every_form or desc_conv_every_form" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_headline.py tests/test_gpu_pipeline.py -x -q --arith pose_conv=winograd43 -k "pose or group_walk or matching_chain or pair_matches or vs_reference or config0" 2>&1 | tail -3
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
for f in sorted(glob.glob("gpurun_out/r4i/kt/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if "wino43v" in name or "cost_l0" in name or "wino_pose" in name:
            n = name.replace("(anonymous namespace)::", "").replace("void ", "")
            print("%-60s %5d %10.1f us" % (n[:n.find("(")], calls, avg / 1e3 if avg > 1e5 else avg))
PY
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -size +20M -delete
