#!/bin/bash
# round 3: convolution kernel variants -- parity (layer / stack tests) + per-kernel time + rate
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3x; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_stages.py -x -q 2>&1 | tail -2
BX_DESC_CONV=winograd timeout 600 python -m pytest tests/test_gpu_stages.py -x -q -k "desc" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 > $OUT/bench_kt.log 2>&1
python - <<'PY'
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/r3x/kt/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if "wino" in name or "conv_kernel" in name or "cost_l0" in name or "patch" in name or "desc_head" in name:
            n = name.replace("(anonymous namespace)::", "").replace("void ", "")
            print("%-60s %5d %10.1f us" % (n[:n.find("(")], calls, avg / 1e3 if avg > 1e5 else avg))
PY
python bench.py --steps 24 --warmup 8 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 > $OUT/bench.json 2> $OUT/bench.err; python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d["stages_ms_per_pair"]
print("value", d["value"], "desc", s.get("desc_conv"), "pose", s.get("pose_net"), "patch", s.get("patch_features"), "ok", d["registered_ok"])
PY
