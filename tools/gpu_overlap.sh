#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/overlap; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python bench.py --steps 64 --warmup 8 --no-cpu-baseline --latency-tiles 0 > $OUT/bench.log 2>&1
python tools/overlap_report.py $OUT/kt > $OUT/overlap.txt 2>&1; cat $OUT/overlap.txt
find $OUT -name '*.db' -size +30M -delete
