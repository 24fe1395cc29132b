#!/bin/bash
# round 3, call D: fixed / new tests (pose_net forms, conv group walk, tiles x conv32, dist) + cost_l0 wave variants + radius hist
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3d; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q -k "group_walk or config0 or 200k or big or armed" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_dist.py tests/test_gpu_bench.py tests/test_gpu_harness.py tests/test_gpu_model.py -x -q 2>&1 | tail -6
CMD="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --latency-tiles 0"
run() { tag=$1; shift; env "$@" $CMD > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python - $OUT/bench_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline_costnet"]; s = d["stages_ms_per_pair"]
    print(sys.argv[2], "value", d["value"], "pose_net", s.get("pose_net"), "costnet frac", r["frac"], "radius", s.get("radius"), "host_ms", d.get("host_ms_per_pair"), "ok", d["registered_ok"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run ot4 BX_COST_OT=4
run ot8 BX_COST_OT=8
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- env BX_COST_OT=4 python bench.py --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --latency-tiles 0 > $OUT/bench_kt.log 2>&1
python - <<'PY'
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/r3d/kt/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    for kn in ("cost_l0", "radius_hist", "patch_features", "patch_axis", "nn1_kernel", "desc_head"):
        rows = list(db.execute(f"select start, end from kernels where name like '%{kn}%' order by start"))
        d = [(e - s) / 1e3 for s, e in rows]
        if d:
            print("%-14s n=%d mean %.1f us" % (kn, len(d), sum(d) / len(d)))
PY
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -size +20M -delete
