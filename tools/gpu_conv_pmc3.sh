#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/convpmc3; rm -rf $OUT; mkdir -p $OUT
for v in ${VARIANTS:-1 0}; do
export BX_CONV32=$v
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace -d $OUT/p$v -o p -- python tools/bench_stage.py conv --iters 4 > $OUT/pmc_$v.log 2>&1
python tools/pmc_mfma.py $OUT/p$v conv
done
find $OUT -name '*.csv' -size +2M -delete
