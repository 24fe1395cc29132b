#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ballpmc; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace -d $OUT/a -o a -- python tools/bench_stage.py ball --iters 4 > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/b -o b -- python tools/bench_stage.py ball --iters 4 > $OUT/b.log 2>&1
python - <<PY
import sqlite3,glob
for d in ("a","b"):
    for f in glob.glob("gpurun_out/ballpmc/%s/*.db"%d):
        db=sqlite3.connect(f)
        rows={}
        for r in db.execute("select kernel_name,counter_name,avg(value),count(*),avg(duration) from counters_collection where kernel_name like '%ball_query%' group by kernel_name,counter_name"):
            rows.setdefault(r[0].split('(')[0][-28:],{})[r[1]]=(r[2],r[3],r[4])
        for k,v in rows.items():
            print(k, " ".join("%s=%.4g"%(n,x[0]) for n,x in sorted(v.items())), "n=%d dur_us=%.1f"%(list(v.values())[0][1], list(v.values())[0][2]/1e3))
PY
