#!/bin/bash
# tools/gpu.sh -- the ONE script behind every `gpurun` call (rounds 1-4 grew 49 one-shot tools/gpu_r*.sh; they are in the git history).
# Runs on the GPU box from the repository root; everything it writes goes to gpurun_out/<OUT> (merged back by gpurun).
#
#   gpurun --timeout 1500 -- 'bash tools/gpu.sh OUT step [step ...]'          steps run in order, a failing step does not stop the rest
#
# steps (arguments after ':' are comma-separated, no spaces):
#   suite[:PYTEST_K]        pytest -m gpu (optionally -k PYTEST_K) + smoke()
#   tests:FILE[,FILE..]     pytest -m gpu on the given tests/ files (without the tests/ prefix)
#   bench[:TAG[,args...]]   python bench.py <args with ',' -> ' '>  -> bench_TAG.json (+ one summary line)
#   layers[:TAG[,SO]]       tools/bench_conv_layers.py (Desc stack per layer); SO = variants/libbufferx_X.so built by tools/build_variant.sh
#   stage:WHAT[,args...]    tools/bench_stage.py WHAT (ball | patch | conv | all)
#   ubench:NAME[,args...]   hipcc tools/ubench/NAME.hip && run it (with args) -> ubench_NAME.txt
#   profile[:TAG]           rocprofv3 kernel-trace + the PMC passes of the bench command -> prof_TAG/ (summaries for profiles/)
#   kstat:TAG,cmd...        rocprofv3 --kernel-trace --stats of `python <cmd with ',' -> ' '>` -> per-kernel table (top 25)
#   env:VAR=VALUE           export VAR for the following steps (e.g. env:BX_HIP_SO=...)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
NAME=$1; shift
OUT=$PWD/gpurun_out/$NAME; mkdir -p $OUT
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d.get("stages_ms_per_pair", {})
    print("value", d["value"], "ms/step", d["ms_per_step"], "p50", d["p50_ms_per_pair"], "ok", d["registered_ok"])
    print("stages", s)
    r, c, n = d.get("roofline") or {}, d.get("roofline_costnet") or {}, d.get("roofline_neighbour_gather") or {}
    print("roofline", r.get("frac"), r.get("avg_launch_ms"), "costnet", c.get("frac"), c.get("avg_launch_ms"), "ng", n.get("frac"), n.get("query_kernel_frac"), "fps", d.get("fps", {}).get("us_per_iteration"))
except Exception as e:
    print("bench line unreadable:", e)
PY
}
for step in "$@"; do
  what=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== $step"
  case $what in
    env) export "$arg";;
    suite)
      if [ -n "$arg" ]; then timeout 3000 python -m pytest tests -x -q -m gpu -k "$arg" 2>&1 | tail -8
      else timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8; fi
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2;;
    tests)
      files=$(echo "$arg" | tr ',' '\n' | sed 's#^#tests/#' | tr '\n' ' ')
      timeout 3000 python -m pytest $files -x -q -m gpu 2>&1 | tail -8;;
    bench)
      tag=${arg%%,*}; rest=""; [[ "$arg" == *,* ]] && rest=$(echo "${arg#*,}" | tr ',' ' ')
      tag=${tag:-default}
      timeout 900 python bench.py $rest > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || tail -c 600 $OUT/bench_$tag.err
      summ $OUT/bench_$tag.json;;
    layers)
      tag=${arg%%,*}; so=""; [[ "$arg" == *,* ]] && so=${arg#*,}
      tag=${tag:-shipped}
      # SO = variants/lib...so (a variant library) or form=NAME (cfg.arith.desc_conv of the shipped library)
      if [[ "$so" == form=* ]]; then timeout 600 python tools/bench_conv_layers.py --tag $tag --form ${so#form=} 2>&1 | tail -3 | tee -a $OUT/layers.jsonl
      elif [ -n "$so" ]; then BX_HIP_SO=$PWD/buffer-x_amd/csrc/$so timeout 600 python tools/bench_conv_layers.py --tag $tag 2>&1 | tail -3 | tee -a $OUT/layers.jsonl
      else timeout 600 python tools/bench_conv_layers.py --tag $tag 2>&1 | tail -3 | tee -a $OUT/layers.jsonl; fi;;
    stage)
      w=${arg%%,*}; rest=""; [[ "$arg" == *,* ]] && rest=$(echo "${arg#*,}" | tr ',' ' ')
      timeout 900 python tools/bench_stage.py $w $rest 2>&1 | tail -12 | tee -a $OUT/stage_$w.jsonl;;
    ubench)
      # a binary cross-compiled in the build container (git-ignored, travels with the snapshot) saves GPU-box minutes
      ub=${arg%%,*}; rest=""; [[ "$arg" == *,* ]] && rest=$(echo "${arg#*,}" | tr ',' ' ')
      if [ -x tools/ubench/$ub ] && [ tools/ubench/$ub -nt tools/ubench/$ub.hip ]; then UB=tools/ubench/$ub
      else UB=/tmp/ub_$ub; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o $UB tools/ubench/$ub.hip 2>&1 | tail -3; fi
      timeout 600 $UB $rest 2>&1 | tee $OUT/ubench_$ub.txt | tail -120;;
    profile)
      TAG=${arg:-r05}; P=$OUT/prof_$TAG; rm -rf $P; mkdir -p $P
      CMD="python bench.py --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --latency-tiles 0 --e2e-pairs 0"
      # kernel trace + stats (one pair in flight: kernel durations not inflated by overlap); PMC passes each in their own run
      # (FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with sys / hip / hsa tracing)
      rocprofv3 --kernel-trace --stats -d $P/kt -o kt -- $CMD > $P/bench_kt.log 2>&1
      rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/pmc_fetch -o f -- $CMD > $P/bench_f.log 2>&1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/pmc_write -o w -- $CMD > $P/bench_w.log 2>&1
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace -d $P/pmc_mfma -o m -- $CMD > $P/bench_m.log 2>&1
      python tools/summarize_prof.py $P > $P/summary.txt 2>&1
      python tools/pmc_traffic.py $P/summary.txt > $P/pmc_traffic.json 2>&1
      python tools/pmc_mfma_json.py $P/pmc_mfma > $P/mfma_busy.json 2>&1
      python tools/src_sha.py --stamp $P/pmc_traffic.json $P/mfma_busy.json
      $CMD > $P/bench_line.json 2> $P/bench_line.err
      find $P -name '*.csv' -size +3M -delete; find $P -name '*.db' -size +30M -delete
      head -70 $P/summary.txt; grep -E "desc_conv_stack|costnet" $P/mfma_busy.json;;
    kstat)
      tag=${arg%%,*}; rest=$(echo "${arg#*,}" | tr ',' ' ')
      K=$OUT/kstat_$tag; rm -rf $K; mkdir -p $K
      timeout 600 rocprofv3 --kernel-trace --stats -d $K/kt -o kt -- python $rest > $K/out.log 2>&1 < /dev/null
      python tools/summarize_prof.py $K 2>/dev/null | head -27 | tee $K/summary.txt
      find $K -name '*.csv' -size +3M -delete; find $K -name '*.db' -size +30M -delete;;
    *) echo "unknown step $what";;
  esac
done
