#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT/prof_conv
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "conv or desc or pose" 2>&1 | tail -3
for gh in ${GHS:-1 0}; do
export BX_CONV_PERSIST=$gh
echo "== BX_CONV_PERSIST=$gh"
rm -rf $OUT/prof_conv/g$gh
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_conv/g$gh -o kt -- python tools/bench_stage.py conv --iters 12 > $OUT/prof_conv/bench_$gh.log 2>&1
grep '"stage"' $OUT/prof_conv/bench_$gh.log
python - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/prof_conv/g$gh/kt_results.db")
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if "conv_kernel" in name or "desc_head" in name: print("%-64s %5d %10.2f us" % (name.replace("(anonymous namespace)::","")[:64], calls, avg))
PY
done
find $OUT/prof_conv -name '*.csv' -size +2M -delete; find $OUT/prof_conv -name '*.db' -size +20M -delete
