#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT/prof_conv
rm -rf $OUT/prof_conv/pmc*
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/prof_conv/pmc1 -o p -- python tools/bench_stage.py conv --iters 4 > $OUT/prof_conv/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/prof_conv/pmc2 -o p -- python tools/bench_stage.py conv --iters 4 > $OUT/prof_conv/pmc2.log 2>&1
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob("gpurun_out/prof_conv/pmc*/*.db")):
    db = sqlite3.connect(f)
    print("==", f)
    try:
        q = "select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"
        rows = list(db.execute(q))
    except Exception as e:
        print("ERR", e); print([r[0] for r in db.execute("select name from sqlite_master")][:40]); continue
    ks = {}
    for k, c, v, n, d in rows:
        if "conv_kernel" not in k: continue
        ks.setdefault(k.replace("(anonymous namespace)::","")[:60], {"dur_us": (d or 0)/1e3})[c] = v
    for k, d in ks.items():
        print(k, {a: round(b, 1) for a, b in d.items()})
PY
tail -3 $OUT/prof_conv/pmc2.log
find $OUT/prof_conv -name '*.csv' -size +2M -delete; find $OUT/prof_conv -name '*.db' -size +20M -delete
