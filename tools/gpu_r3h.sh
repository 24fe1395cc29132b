#!/bin/bash
# round 3, call H: pipelined Winograd kernel -- parity in Winograd mode + rate against the direct form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3h; rm -rf $OUT; mkdir -p $OUT
export BX_DESC_CONV=winograd   # the default since round 3
timeout 600 python -m pytest tests/test_gpu_stages.py -x -q -k "desc_conv_layer or desc_net" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_headline.py -x -q -k "group_walk or descriptor_chain or vs_reference" 2>&1 | tail -4
CMD="python bench.py --steps 16 --warmup 4 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0"
run() { tag=$1; shift; env "$@" $CMD > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python - $OUT/bench_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]; s = d["stages_ms_per_pair"]
    print(sys.argv[2], "value", d["value"], "desc_conv ms/pair", s.get("desc_conv"), "stack ms", r["avg_launch_ms"], "ok", d["registered_ok"], "p50@1", d["p50_ms_per_pair_inflight1"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run winograd BX_DESC_CONV=winograd
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 > $OUT/bench_kt.log 2>&1
python - <<'PY'
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/r3h/kt/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if "wino" in name:
            n = name.replace("(anonymous namespace)::", "").replace("void ", "")
            print("%-60s %5d %10.1f us" % (n[:n.find("(")], calls, avg / 1e3 if avg > 1e5 else avg))
PY
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -size +20M -delete
