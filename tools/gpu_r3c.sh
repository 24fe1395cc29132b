#!/bin/bash
# round 3, call C: collapsed CostNet layer 0 (k_cost.hip) + real-size parity (tests/test_gpu_headline.py on the reference-minted fixtures)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3c; rm -rf $OUT; mkdir -p $OUT
export BX_REALSIZE_REPORT=$OUT/realsize_report.jsonl
timeout 600 python -m pytest tests/test_gpu_stages.py -x -q -k "pose_net or ball" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_headline.py -x -q -s 2>&1 | grep -v "^$" | tail -15
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_tiles.py tests/test_gpu_kiss.py -x -q 2>&1 | tail -5
CMD="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --latency-tiles 0"
run() { tag=$1; shift; env "$@" $CMD > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python - $OUT/bench_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline_costnet"]; s = d["stages_ms_per_pair"]
    print(sys.argv[2], "value", d["value"], "pose_net ms/pair", s.get("pose_net"), "costnet frac", r["frac"], "ms/launch", r["avg_launch_ms"], "desc", s.get("desc_conv"), "ok", d["registered_ok"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run collapsed BX_X=0
run direct BX_COST_L0=direct
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --latency-tiles 0 > $OUT/bench_kt.log 2>&1
python - <<'PY'
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/r3c/kt/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    for kn in ("cost_l0", "cost_l1", "2, 27, 972"):
        rows = list(db.execute(f"select start, end from kernels where name like '%{kn}%' order by start"))
        d = [(e - s) / 1e3 for s, e in rows]
        if d:
            print("%-14s n=%d mean %.1f us" % (kn, len(d), sum(d) / len(d)))
PY
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -size +20M -delete
