#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench command + PMC passes (HBM traffic, MFMA occupancy).
# Outputs land in gpurun_out/prof_<tag>; summarize_prof.py / pmc_traffic.py / pmc_mfma_json.py turn them into the files under profiles/
# (copy them there afterwards); every derived JSON is stamped with the sha256 of the HIP sources it was measured on (tools/src_sha.py).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=${1:-r04}
rm -rf $OUT/prof_$TAG && mkdir -p $OUT/prof_$TAG
CMD="python bench.py --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0"
# (1) kernel trace + stats (1 pair in flight so kernel durations are not inflated by overlap)
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/kt -o kt -- $CMD > $OUT/prof_$TAG/bench_kt.log 2>&1
# (2) PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE cannot share a pass)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_$TAG/pmc_fetch -o f -- $CMD > $OUT/prof_$TAG/bench_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_$TAG/pmc_write -o w -- $CMD > $OUT/prof_$TAG/bench_w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace -d $OUT/prof_$TAG/pmc_mfma -o m -- $CMD > $OUT/prof_$TAG/bench_m.log 2>&1
python tools/summarize_prof.py $OUT/prof_$TAG > $OUT/prof_$TAG/summary.txt 2>&1
python tools/pmc_traffic.py $OUT/prof_$TAG/summary.txt > $OUT/prof_$TAG/pmc_traffic.json 2>&1
python tools/pmc_mfma_json.py $OUT/prof_$TAG/pmc_mfma > $OUT/prof_$TAG/mfma_busy.json 2>&1
python tools/src_sha.py --stamp $OUT/prof_$TAG/pmc_traffic.json $OUT/prof_$TAG/mfma_busy.json
# the same command, un-profiled, for the bench line that goes with the profile; then the default command (what the driver runs)
$CMD > $OUT/prof_$TAG/bench_line.json 2> $OUT/prof_$TAG/bench_line.err
python bench.py > $OUT/prof_$TAG/bench_default.json 2> $OUT/prof_$TAG/bench_default.err
# keep only the small artefacts
find $OUT/prof_$TAG -name '*.csv' -size +3M -delete
find $OUT/prof_$TAG -name '*.db' -size +30M -delete
head -60 $OUT/prof_$TAG/summary.txt; grep -E "desc_conv_stack|costnet" $OUT/prof_$TAG/mfma_busy.json
