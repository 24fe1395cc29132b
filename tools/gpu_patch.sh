#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT/prof_patch
timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pipeline.py -x -q -k "patch or pipeline or register" 2>&1 | tail -3
rm -rf $OUT/prof_patch/x
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_patch/x -o kt -- python tools/bench_stage.py patch --iters 10 > $OUT/prof_patch/x.log 2>&1
grep '"stage"' $OUT/prof_patch/x.log
python - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/prof_patch/x/kt_results.db")
rows=list(db.execute("select start, end from kernels where name like '%patch_features%' order by start"))
d=[(e-s)/1e3 for s,e in rows]
print("patch_features per case (3 scales x not-aligned/aligned):", ["%.0f" % (sum(d[i*13+3:(i+1)*13])/10) for i in range(6)])
PY
rm -rf $OUT/prof_patch/x
