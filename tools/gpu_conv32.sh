#!/bin/bash
# A/B of the Cylindrical_Net stack: 32x32x2 kernels (k_conv32.hip) vs the 16x16x4 kernels (k_conv.hip), per-layer durations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/conv32; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_headline.py -x -q -k "conv or desc or walk or descriptor_chain" 2>&1 | tail -3
for v in ${VARIANTS:-1 0}; do
export BX_CONV32=$v
echo "== BX_CONV32=$v"
rm -rf $OUT/g
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/g -o kt -- python tools/bench_stage.py conv --iters 12 > $OUT/bench_$v.log 2>&1
grep '"stage"' $OUT/bench_$v.log
python - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/conv32/g/kt_results.db")
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if "conv" in name or "desc_head" in name: print("%-64s %5d %10.1f us" % (name.replace("(anonymous namespace)::","").replace("void ","")[:64], calls, avg))
PY
done
rm -rf $OUT/g
BX_BALL_DEBUG=1 python tools/bench_stage.py conv --iters 8 2>&1 | grep -A1 "conv32 layer-3"
