#!/bin/bash
# PMC pass over patch_features_kernel (VALU / LDS activity, waits) -- planning data for the wave-imbalance item
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/patchpmc; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 70 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/a -o a -- python tools/bench_stage.py patch --n 38000 --iters 3 > $OUT/a.log 2>&1
python - <<PY
import sqlite3,glob
for f in glob.glob("gpurun_out/patchpmc/a/*.db"):
    db=sqlite3.connect(f)
    rows={}
    for r in db.execute("select kernel_name,counter_name,avg(value),count(*),avg(duration) from counters_collection where kernel_name like '%patch_features%' group by kernel_name,counter_name"):
        rows.setdefault(r[0].split('(')[0][-30:],{})[r[1]]=(r[2],r[3],r[4])
    for k,v in rows.items():
        print(k, " ".join("%s=%.4g"%(n,x[0]) for n,x in sorted(v.items())), "n=%d dur_us=%.1f"%(list(v.values())[0][1], list(v.values())[0][2]/1e3))
PY
