#!/bin/bash
# GPU box: parity of the neighbour-gather stage + rocprofv3 kernel trace of the stage micro-benchmark
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
rm -rf $OUT/prof_ball; mkdir -p $OUT/prof_ball
for div in ${WAVES:-1 2 4}; do
export BX_BALL_WAVES=$div
echo "=== BX_BALL_WAVES=$div"
timeout 600 python -m pytest tests/test_gpu_stages.py -x -q -k "ball" 2>&1 | tail -3
for n in ${NS:-30000 60000}; do
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_ball/d${div}_n$n -o kt -- python tools/bench_stage.py ball --n $n > $OUT/prof_ball/bench_d${div}_$n.log 2>&1
grep '"stage"' $OUT/prof_ball/bench_d${div}_$n.log | grep '"idx": true'
done
done
python - <<'PY'
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/prof_ball/*/*.db")):
    db = sqlite3.connect(f); print("==", f)
    for kn in ("ball_query","cell_count","bbox_kernel","cell_scatter","scan_apply","scan_sums","grid_setup","fillBuffer"):
        rows=list(db.execute(f"select start, end from kernels where name like '%{kn}%' order by start"))
        d=[(e-s)/1e3 for s,e in rows]
        if len(d) >= 138:
            d = d[-138:]
            print("%-14s" % kn, " ".join("%6.1f" % (sum(d[i*23+3:(i+1)*23])/20) for i in range(6)), "us  (3 scales x idx/no-idx)")
PY
find $OUT/prof_ball -name '*.csv' -size +2M -delete; find $OUT/prof_ball -name '*.db' -size +20M -delete
