#!/bin/bash
# round 4: radius histogram with 16-bit LDS counters: parity (incl. the concentrated-bin cases), then the bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4s; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pipeline.py -x -q -k "radius or pipeline or register" 2>&1 | tail -3
timeout 600 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --e2e-pairs 0 --latency-tiles 0 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "stages", d["stages_ms_per_pair"]["radius"])
PY
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 > $OUT/kt.log 2>&1
python - <<'PY'
import sqlite3, glob
for f in glob.glob("gpurun_out/r4s/kt/*.db"):
    db = sqlite3.connect(f)
    for n, c, a in db.execute("select k.name, count(*), avg(end-start)/1e3 from kernels k where k.name like '%radius%' group by k.name"):
        print(n[:60], c, round(a, 1), "us")
PY
find $OUT -name '*.db' -size +20M -delete
