#!/bin/bash
# round 3: the whole GPU suite (default = Winograd Desc layers) + the direct form on the chain tests + smoke + the r03 profile set
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
export BX_REALSIZE_REPORT=$PWD/gpurun_out/r3f/realsize_report.jsonl
rm -f $BX_REALSIZE_REPORT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
BX_REALSIZE_REPORT=$PWD/gpurun_out/r3f/realsize_report_direct.jsonl BX_DESC_CONV=direct timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_headline.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/profile_r3.sh r03 2>&1 | tail -70
BX_DESC_CONV=direct python bench.py --steps 24 --warmup 8 --no-cpu-baseline --e2e-pairs 0 > gpurun_out/r3f/bench_direct.json 2> gpurun_out/r3f/bench_direct.err
