#!/bin/bash
# MFMA-busy / effective-clock counters of the Cylindrical_Net kernels, both variants; plus zero-operand and non-persistent timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/convpmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -iE "mfma|GUI_ACTIVE|SQ_BUSY" | head -20 > $OUT/counters.txt
for v in 1 0; do
export BX_CONV32=$v
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/p$v -o p -- python tools/bench_stage.py conv --iters 4 > $OUT/pmc_$v.log 2>&1
python tools/pmc_mfma.py $OUT/p$v conv
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/q$v -o q -- python tools/bench_stage.py conv --iters 4 > $OUT/pmcq_$v.log 2>&1
python tools/pmc_mfma.py $OUT/q$v conv 2>&1 | head -3
python - <<PY
import sqlite3,glob
for f in glob.glob("gpurun_out/convpmc/q$v/*.db"):
    db=sqlite3.connect(f)
    for r in db.execute("select kernel_name,counter_name,avg(value) from counters_collection where kernel_name like '%conv%' group by kernel_name,counter_name"):
        print(r[0][:60].replace("(anonymous namespace)::",""), r[1], "%.4g"%r[2])
PY
done
for v in 1 0; do for z in "" 1; do for ps in 1 0; do
echo "== BX_CONV32=$v BX_BENCH_ZERO=$z BX_CONV_PERSIST=$ps"
BX_CONV32=$v BX_BENCH_ZERO=$z BX_CONV_PERSIST=$ps timeout 200 python tools/bench_stage.py conv --iters 24 2>&1 | grep '"stage"'
done; done; done
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -size +20M -delete
