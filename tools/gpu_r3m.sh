#!/bin/bash
# round 3, call M: split-precision measurement (BX_EXP_SPLIT_CONV=1, k_split.hip; measurement only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3m; rm -rf $OUT; mkdir -p $OUT
BX_SPLIT_REPORT=$OUT/split_report.jsonl timeout 900 python -m pytest tests/test_gpu_split_precision.py -q 2>&1 | tail -15
BX_EXP_SPLIT_CONV=1 BX_REALSIZE_REPORT=$OUT/realsize_report_split.jsonl timeout 900 python -m pytest tests/test_gpu_headline.py -q -k vs_reference 2>&1 | tail -8
cat $OUT/split_report.jsonl | cut -c1-700
cat $OUT/realsize_report_split.jsonl | cut -c1-900
CMD="python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0"
BX_EXP_SPLIT_CONV=1 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $CMD > $OUT/bench_kt.log 2>&1
python - <<'PY'
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/r3m/kt/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if "wino_kernel" in name or "split" in name:
            n = name.replace("(anonymous namespace)::", "").replace("void ", "")
            print("%-60s %5d %10.1f us" % (n[:n.find("(")], calls, avg / 1e3 if avg > 1e5 else avg))
PY
BX_EXP_SPLIT_CONV=1 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 > $OUT/bench_split.json 2> $OUT/bench_split.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3m/bench_split.json").read().strip().splitlines()[-1])
print("split bench (NOT a headline): value", d["value"], "ok", d["registered_ok"], "desc", d["stages_ms_per_pair"]["desc_conv"])
PY
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -size +20M -delete
