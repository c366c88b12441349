#!/bin/bash
# End of round 5, after the conv2 kernels were rebuilt: the whole GPU suite, the driver's bench command (without the minutes-long CPU
# leg: profiles/r05z_bench_driver_cmd_with_cpu_leg.json holds it), kernel stats of configs 3 and 5a.
#   gpurun --timeout 1100 -- 'bash tools/r05_final.sh r05f'
tag=${1:-r05f}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
here=$PWD
timeout 800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -n 5 $out/pytest_gpu.log
timeout 60 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.log
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_cmd_no_cpu_leg.json 2> $out/bench_driver_cmd.err; echo "bench rc=$?"; grep -o '"ms_per_step": [0-9.]*' $out/bench_driver_cmd_no_cpu_leg.json | head -1
cd /tmp
for cfg in cfg3 cfg5a; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $here/$out/prof_$cfg -o $cfg -- python $here/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-stock-baseline --no-other-configs > $here/$out/prof_$cfg.log 2>&1; echo "rocprof $cfg rc=$?"
  db=$(find $here/$out/prof_$cfg -name "*.db" | head -n 1)
  [ -n "$db" ] && python $here/tools/rocpd_stats.py $db > $here/$out/kernel_stats_$cfg.md 2>> $here/$out/prof_$cfg.log
  head -n 10 $here/$out/kernel_stats_$cfg.md | cut -c1-160
  find $here/$out/prof_$cfg -size +20M -delete
done
