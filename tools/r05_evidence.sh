#!/bin/bash
# Round 5 end-of-round evidence from ONE build in ONE session:
#   the driver's own bench command (stock + other configurations + CPU legs), rocprofv3 kernel stats per configuration, PMC passes
#   (separate, kernel-trace only) for cfg3, the data-parallel fields on one GPU (DS2_FORCE_DDP=1), chunked inference.
#   gpurun --timeout 2400 -- 'bash tools/r05_evidence.sh r05z'      (BENCH_EXTRA=--no-cpu-baseline skips the minutes-long CPU leg)
tag=${1:-r05z}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
here=$PWD
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 $BENCH_EXTRA > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err; echo "bench (driver command) rc=$?"; tail -c 400 $out/bench_driver_cmd.json; echo; tail -n 12 $out/bench_driver_cmd.err
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-stock-baseline > $out/bench_200steps.json 2> $out/bench_200steps.err; echo "bench 200 steps rc=$?"; grep -o '"ms_per_step": [0-9.]*' $out/bench_200steps.json
DS2_FORCE_DDP=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock-baseline > $out/bench_force_ddp.json 2> $out/bench_force_ddp.err; echo "bench DS2_FORCE_DDP=1 rc=$?"; python -c "
import json,sys
j=json.loads([l for l in open('$out/bench_force_ddp.json') if l.startswith('{')][-1]); print(j['ms_per_step'], json.dumps(j.get('data_parallel'))[:600])"
cd /tmp
for cfg in cfg3 cfg2 cfg5a cfg5b; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $here/$out/prof_$cfg -o $cfg -- python $here/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-stock-baseline > $here/$out/prof_$cfg.log 2>&1; echo "rocprof $cfg rc=$?"
  db=$(find $here/$out/prof_$cfg -name "*.db" | head -n 1)
  [ -n "$db" ] && python $here/tools/rocpd_stats.py $db > $here/$out/kernel_stats_$cfg.md 2>> $here/$out/prof_$cfg.log
  head -n 8 $here/$out/kernel_stats_$cfg.md | cut -c1-160
  find $here/$out/prof_$cfg -size +20M -delete
done
IFS=';' read -ra sets <<< "FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
i=0
for ctrs in "${sets[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $here/$out/pmc$i -o cfg3 -- python $here/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-stock-baseline > $here/$out/pmc$i.log 2>&1; echo "pmc [$ctrs] rc=$?"
  db=$(find $here/$out/pmc$i -name "*.db" | head -n 1)
  [ -n "$db" ] && python $here/tools/rocpd_pmc.py $db > $here/$out/pmc$i.md 2>> $here/$out/pmc$i.log
  head -n 6 $here/$out/pmc$i.md | cut -c1-160
  find $here/$out/pmc$i -size +20M -delete
done
cd $here
(timeout 200 python tools/bench_stream.py; timeout 200 python tools/bench_stream.py --stock) > $out/stream_inference.jsonl 2> $out/stream_inference.err; echo "stream rc=$?"; tail -n 4 $out/stream_inference.jsonl | cut -c1-300
