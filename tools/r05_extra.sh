tag=r05g; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; here=$PWD
timeout 120 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-stock-baseline > $out/bench_200steps.json 2> $out/bench_200steps.err; grep -o '"ms_per_step": [0-9.]*' $out/bench_200steps.json | head -1
cd /tmp
for cfg in cfg5b cfg2; do
  timeout 150 rocprofv3 --kernel-trace --stats -d $here/$out/prof_$cfg -o $cfg -- python $here/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-stock-baseline --no-other-configs > $here/$out/prof_$cfg.log 2>&1; echo "rocprof $cfg rc=$?"
  db=$(find $here/$out/prof_$cfg -name "*.db" | head -n 1)
  [ -n "$db" ] && python $here/tools/rocpd_stats.py $db > $here/$out/kernel_stats_$cfg.md 2>> $here/$out/prof_$cfg.log
  head -n 6 $here/$out/kernel_stats_$cfg.md | cut -c1-150
  find $here/$out/prof_$cfg -size +20M -delete
done
