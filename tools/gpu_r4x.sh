#!/bin/bash
# round 4: SQ counters of the shipped F(4x4) kernels (per layer), two passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4x; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace -d $OUT/pmc1 -o p -- python tools/bench_conv_layers.py --K 5000 --iters 3 > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_MISC --kernel-trace -d $OUT/pmc2 -o p -- python tools/bench_conv_layers.py --K 5000 --iters 3 > $OUT/pmc2.log 2>&1
python - <<'PY' | tee gpurun_out/r4x/counters.txt
import sqlite3, glob
for f in sorted(glob.glob("gpurun_out/r4x/pmc*/*.db")):
    db = sqlite3.connect(f)
    print("==", f)
    try:
        q = "select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"
        rows = list(db.execute(q))
    except Exception as e:
        print("ERR", e); print([r[0] for r in db.execute("select name from sqlite_master")][:40]); continue
    ks = {}
    for k, c, v, n, d in rows:
        if "wino43_kernel" not in k: continue
        ks.setdefault(k.replace("(anonymous namespace)::","")[:60], {"dur_us": (d or 0)/1e3})[c] = v
    for k, d in ks.items():
        print(k, {a: round(b, 1) for a, b in d.items()})
PY
tail -2 $OUT/pmc2.log
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -size +20M -delete
