#!/bin/bash
# round 4, call B: v3 = slab + one-channel transform + swapped-operand output; start-stagger sweep (BX_W43_STAG_A: cycles x NCHUNK for
# workgroups with slack in the ragged last round, _B: for the others)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4b; rm -rf $OUT; mkdir -p $OUT
V=$PWD/buffer-x_amd/csrc/variants
echo "== parity v3"
BX_HIP_SO=$V/libbufferx_v3.so timeout 300 python -m pytest tests/test_gpu_stages.py -x -q -k "desc_conv_layer_exact or desc_net" 2>&1 | tail -2
BX_HIP_SO=$V/libbufferx_v3.so timeout 400 python -m pytest tests/test_gpu_headline.py -x -q -k "group_walk" 2>&1 | tail -2
run() { v=$1; a=$2; b=$3; BX_W43_STAG_A=$a BX_W43_STAG_B=$b BX_HIP_SO=$V/libbufferx_$v.so timeout 200 python tools/bench_conv_layers.py --tag "$v,A=$a,B=$b" 2>&1 | tail -1 | tee -a $OUT/layers.jsonl; }
run v3 0 0
BX_W43_STAMPS=1 run v3s 0 0
run v3 6000 0
run v3 12000 0
BX_W43_STAMPS=1 run v3s 12000 0
run v3 12000 3000
run v3 12000 6000
run v3 24000 0
run v2 12000 0
run v2 12000 6000
for ab in "0 0" "12000 0"; do set -- $ab
  BX_W43_STAG_A=$1 BX_W43_STAG_B=$2 BX_HIP_SO=$V/libbufferx_v3.so timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0 > $OUT/bench_v3_$1.json 2> $OUT/bench_v3_$1.err
  python - $OUT/bench_v3_$1.json "v3 A=$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d["stages_ms_per_pair"]
    print(sys.argv[2], "value", d["value"], "desc", s.get("desc_conv"), "pose", s.get("pose_net"), "ok", d["registered_ok"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
