#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command + PMC passes.
# Outputs land in gpurun_out/prof_* ; the summaries worth judging are copied to profiles/ by hand.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=${1:-r01}
rm -rf $OUT/prof_$TAG && mkdir -p $OUT/prof_$TAG
# (1) kernel trace + stats of the bench command (1 pair in flight so kernel durations are not inflated by overlap)
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/kt -o kt -- python bench.py --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/prof_$TAG/bench_kt.log 2>&1
# (2) PMC passes (separate runs; FETCH_SIZE and WRITE_SIZE cannot share a pass)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_$TAG/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/prof_$TAG/bench_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_$TAG/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/prof_$TAG/bench_w.log 2>&1
python tools/summarize_prof.py $OUT/prof_$TAG > $OUT/prof_$TAG/summary.txt 2>&1
# keep only the small artefacts
find $OUT/prof_$TAG -name '*.csv' -size +3M -delete
ls -laR $OUT/prof_$TAG | head -50
