#!/bin/bash
# round 3: the whole GPU suite (default = Winograd F(4x4, 3x3) Desc layers) + the F(2x2, 3x3) and direct forms on the chain tests + smoke +
# the r03 profile set
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
export BX_REALSIZE_REPORT=$PWD/gpurun_out/r3f/realsize_report.jsonl
rm -f $BX_REALSIZE_REPORT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
BX_REALSIZE_REPORT=$PWD/gpurun_out/r3f/realsize_report_direct.jsonl BX_DESC_CONV=direct timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_headline.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -4
BX_REALSIZE_REPORT=$PWD/gpurun_out/r3f/realsize_report_wino22.jsonl BX_DESC_CONV=winograd timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_headline.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
BX_POSE_CONV=direct timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pipeline.py -m gpu -x -q -k "pose or golden or pipeline" 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/profile_r3.sh r03 2>&1 | tail -70
BX_DESC_CONV=direct python bench.py --steps 24 --warmup 8 --no-cpu-baseline --e2e-pairs 0 > gpurun_out/r3f/bench_direct.json 2> gpurun_out/r3f/bench_direct.err
BX_DESC_CONV=winograd python bench.py --steps 24 --warmup 8 --no-cpu-baseline --e2e-pairs 0 > gpurun_out/r3f/bench_wino22.json 2> gpurun_out/r3f/bench_wino22.err
for w in kitti tiers 3dlomatch; do python bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline --e2e-pairs 0 --latency-tiles 0 > gpurun_out/r3f/bench_$w.json 2> gpurun_out/r3f/bench_$w.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3f/bench_*.json")) + ["gpurun_out/prof_r03/bench_default.json", "gpurun_out/prof_r03/bench_line.json"]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d.get("registered_ok"), d["config"].get("workload"), "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"))
    except Exception as e:
        print(f, "FAILED", e)
PY
