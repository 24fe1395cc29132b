#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT/prof_conv
for v in $VARIANTS; do
for gh in 0; do
export BX_CONV_GHALF=$gh
export BX_HIP_SO=$PWD/buffer-x_amd/csrc/_exp/libbx_$v.so
echo "== $v GHALF=$gh"
rm -rf $OUT/prof_conv/x
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_conv/x -o kt -- python tools/bench_stage.py conv --iters 8 > $OUT/prof_conv/x.log 2>&1
grep '"stage"' $OUT/prof_conv/x.log
python - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/prof_conv/x/kt_results.db")
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if "conv_kernel" in name: print("%-64s %5d %10.2f us" % (name.replace("(anonymous namespace)::","")[:64], calls, avg))
PY
done
done
rm -rf $OUT/prof_conv/x
