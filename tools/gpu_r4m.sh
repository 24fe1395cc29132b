#!/bin/bash
# round 4, call M: the GPU suite under the non-default arithmetic forms (pytest --arith): every form against its own restatement + the fixtures
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
T="tests/test_gpu_stages.py tests/test_gpu_headline.py tests/test_gpu_pipeline.py tests/test_gpu_degenerate.py tests/test_gpu_tiles.py"
for ar in "desc_conv=direct,pose_conv=direct,cost_l0=direct" "desc_conv=winograd22,pose_conv=winograd22"; do
  echo "== --arith $ar"
  timeout 1500 python -m pytest $T -x -q --arith $ar 2>&1 | tail -2
done
