#!/bin/bash
# One GPU-box session: parity tests, then the bench configurations (+ optional rocprofv3 kernel stats of the cfg3 bench).
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag> [pytest-args...]'
#   env: CONFIGS="cfg3 cfg2" (default all four), PROFILE=1 (rocprofv3 --kernel-trace --stats of 3 cfg3 steps), EXTRA="cmd" (run last)
tag=${1:-run}; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1000 python -m pytest tests -m gpu -q -s --maxfail=25 --tb=short "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
grep -E "passed|failed|Error|FAILED" $out/pytest.log | tail -n 30
for cfg in ${CONFIGS:-cfg3 cfg2 cfg5a cfg5b}; do
  steps=20; [ $cfg != cfg3 ] && steps=6
  timeout 300 python bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-stock-baseline $BENCH_ARGS > $out/bench_$cfg.json 2> $out/bench_$cfg.err; echo "bench $cfg rc=$?"
  tail -c 1200 $out/bench_$cfg.json; echo
done
if [ -n "$PROFILE" ]; then
  here=$PWD; cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $here/$out/prof -o cfg3 -- python $here/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stock-baseline > $here/$out/prof.log 2>&1; echo "rocprof rc=$?"
  cd $here
  db=$(find $out/prof -name "*.db" | head -n 1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py $db > $out/kernel_stats_cfg3.md 2>> $out/prof.log; fi
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_cfg3.csv \;
  head -n 24 $out/kernel_stats_cfg3.md
  find $out/prof -size +20M -delete
fi
if [ -n "$PMC" ]; then   # PMC counters: separate passes, kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section)
  here=$PWD; cd /tmp
  IFS=';' read -ra sets <<< "${PMC_SETS:-FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE}"
  i=0
  for ctrs in "${sets[@]}"; do
    i=$((i+1)); tag2=pmc$i
    timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d $here/$out/$tag2 -o cfg3 -- python $here/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-stock-baseline > $here/$out/$tag2.log 2>&1; echo "pmc [$ctrs] rc=$?"
    db=$(find $here/$out/$tag2 -name "*.db" | head -n 1)
    [ -n "$db" ] && python $here/tools/rocpd_pmc.py $db > $here/$out/$tag2.md 2>> $here/$out/$tag2.log
    head -n 8 $here/$out/$tag2.md
    find $here/$out/$tag2 -size +20M -delete
  done
  cd $here
fi
if [ -n "$EXTRA" ]; then bash -c "$EXTRA" > $out/extra.log 2>&1; echo "extra rc=$?"; tail -n 40 $out/extra.log; fi
