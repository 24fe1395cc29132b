#!/bin/bash
# GPU box: dynamic instruction counts of the neighbour-gather query kernel (rocprofv3 PMC, own pass, kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT/prof_ball
rm -rf $OUT/prof_ball/pmc*
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d $OUT/prof_ball/pmc1 -o p -- python tools/bench_stage.py ball --n ${N:-30000} --iters 3 > $OUT/prof_ball/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/prof_ball/pmc2 -o p -- python tools/bench_stage.py ball --n ${N:-30000} --iters 3 > $OUT/prof_ball/pmc2.log 2>&1
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob("gpurun_out/prof_ball/pmc*/*.db")):
    db = sqlite3.connect(f)
    print("==", f)
    try:
        rows = list(db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print("ERR", e); continue
    ks = {}
    for k, c, v, n, d in rows:
        if "ball_query" not in k: continue
        ks.setdefault(k.replace("(anonymous namespace)::","")[:40], {"n": n, "dur_us": (d or 0)/1e3})[c] = v
    for k, d in ks.items():
        print(k, {a: round(b, 1) for a, b in d.items()})
PY
tail -2 $OUT/prof_ball/pmc2.log
find $OUT/prof_ball -name '*.csv' -size +2M -delete; find $OUT/prof_ball -name '*.db' -size +20M -delete
