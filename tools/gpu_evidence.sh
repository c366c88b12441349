#!/bin/bash
# One GPU-box session that collects the round's evidence from ONE build: default bench line, the other configurations, stock
# PyTorch-ROCm lines, rocprofv3 kernel stats per configuration, PMC passes (separate, kernel-trace only) for cfg3.
#   gpurun --timeout 1500 -- 'bash tools/gpu_evidence.sh <tag>'        env: SKIP_DEFAULT=1, SKIP_STOCK=1, SKIP_PMC=1, SKIP_STREAM=1
tag=${1:-ev}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
here=$PWD
if [ -z "$SKIP_DEFAULT" ]; then   # the default line includes the CPU leg (minutes): the end-of-round gate (tools/r06_gate.sh) runs the driver's command instead
  timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?"; tail -c 600 $out/bench_default.json; echo
fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock-baseline --no-other-configs > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "bench cfg3 rc=$?"; grep -o '"ms_per_step": [0-9.]*' $out/bench_cfg3.json | head -1
for cfg in cfg2 cfg5a cfg5b; do
  timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-stock-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err; echo "bench $cfg rc=$?"
  grep -o '"ms_per_step": [0-9.]*' $out/bench_$cfg.json
done
if [ -z "$SKIP_STOCK" ]; then
  for cfg in cfg3 cfg2 cfg5a cfg5b; do
    timeout 400 python bench.py --stock --config $cfg --steps 3 --warmup 2 > $out/bench_stock_$cfg.json 2> $out/bench_stock_$cfg.err; echo "stock $cfg rc=$?"
    grep -o '"ms_per_step": [0-9.]*' $out/bench_stock_$cfg.json
  done
fi
cd /tmp
for cfg in cfg3 cfg2 cfg5a cfg5b; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $here/$out/prof_$cfg -o $cfg -- python $here/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-stock-baseline > $here/$out/prof_$cfg.log 2>&1; echo "rocprof $cfg rc=$?"
  db=$(find $here/$out/prof_$cfg -name "*.db" | head -n 1)
  [ -n "$db" ] && python $here/tools/rocpd_stats.py $db > $here/$out/kernel_stats_$cfg.md 2>> $here/$out/prof_$cfg.log
  head -n 8 $here/$out/kernel_stats_$cfg.md | cut -c1-160
  find $here/$out/prof_$cfg -size +20M -delete
done
if [ -z "$SKIP_PMC" ]; then
  IFS=';' read -ra sets <<< "FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
  i=0
  for ctrs in "${sets[@]}"; do
    i=$((i+1))
    timeout 400 rocprofv3 --kernel-trace --pmc $ctrs -d $here/$out/pmc$i -o cfg3 -- python $here/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-stock-baseline > $here/$out/pmc$i.log 2>&1; echo "pmc [$ctrs] rc=$?"
    db=$(find $here/$out/pmc$i -name "*.db" | head -n 1)
    [ -n "$db" ] && python $here/tools/rocpd_pmc.py $db > $here/$out/pmc$i.md 2>> $here/$out/pmc$i.log
    head -n 6 $here/$out/pmc$i.md | cut -c1-160
    find $here/$out/pmc$i -size +20M -delete
  done
fi
cd $here
if [ -z "$SKIP_STREAM" ]; then
  (timeout 200 python tools/bench_stream.py; timeout 200 python tools/bench_stream.py --stock) > $out/stream_inference.jsonl 2> $out/stream_inference.err; echo "stream rc=$?"; tail -n 4 $out/stream_inference.jsonl | cut -c1-300
fi
