#!/bin/bash
# round 3, call P: neighbour-gather query kernel, waves per keypoint (BX_BALL_WAVES) -- per-kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3p; rm -rf $OUT; mkdir -p $OUT
for w in ${BALL_WAVES_LIST:-0 1 2}; do
BX_BALL_WAVES=$w rocprofv3 --kernel-trace --stats -d $OUT/kt$w -o kt -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 > $OUT/bench_kt$w.log 2>&1
python - $w <<'PY'
import glob, sqlite3, sys
w = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/r3p/kt%s/**/*.db" % w, recursive=True)):
    db = sqlite3.connect(f)
    for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if "ball" in name:
            n = name.replace("(anonymous namespace)::", "").replace("void ", "")
            print("waves=%s %-44s %5d %10.1f us" % (w, n[:n.find("(")], calls, avg / 1e3 if avg > 1e5 else avg))
PY
python - $OUT/bench_kt$w.log <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    r = d["roofline_neighbour_gather"]
    print("   stage frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "query_kernel_avg_ms", r["query_kernel_avg_ms"], "ok", d["registered_ok"])
except Exception as e:
    print("   FAILED", e)
PY
done
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -size +20M -delete
