#!/bin/bash
# round 3, call F: the whole GPU suite + smoke + the r03 profile set
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
export BX_REALSIZE_REPORT=$PWD/gpurun_out/r3f/realsize_report.jsonl
rm -f $BX_REALSIZE_REPORT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/profile_r3.sh r03 2>&1 | tail -70
