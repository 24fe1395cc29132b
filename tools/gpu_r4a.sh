#!/bin/bash
# round 4, call A: F(4x4) kernel variants -- v0 round-3 slab kernel, v1 window straight from global memory, v2 = v1 + swapped operands /
# two-round output exchange, v2w = v2 with the window loads spread over the planes, v2s = v2 with phase stamps.  Parity + per-layer time.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4a; rm -rf $OUT; mkdir -p $OUT
V=$PWD/buffer-x_amd/csrc/variants
for v in v2 v1 v2w; do
  echo "== parity $v"
  BX_HIP_SO=$V/libbufferx_$v.so timeout 300 python -m pytest tests/test_gpu_stages.py -x -q -k "desc_conv_layer_exact or desc_net" 2>&1 | tail -3
done
for v in v2 v1; do
  echo "== group walk $v"
  BX_HIP_SO=$V/libbufferx_$v.so timeout 400 python -m pytest tests/test_gpu_headline.py -x -q -k "group_walk" 2>&1 | tail -3
done
for v in v0 v1 v2 v2w; do
  BX_HIP_SO=$V/libbufferx_$v.so timeout 200 python tools/bench_conv_layers.py --tag $v 2>&1 | tail -1 | tee -a $OUT/layers.jsonl
done
BX_W43_STAMPS=1 BX_HIP_SO=$V/libbufferx_v2s.so timeout 200 python tools/bench_conv_layers.py --tag v2s 2>&1 | tail -1 | tee -a $OUT/layers.jsonl
for v in v0 v2; do
  BX_HIP_SO=$V/libbufferx_$v.so timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d["stages_ms_per_pair"]
    print(sys.argv[2], "value", d["value"], "desc", s.get("desc_conv"), "pose", s.get("pose_net"), "ok", d["registered_ok"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
