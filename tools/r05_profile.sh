#!/bin/bash
# rocprofv3 kernel stats of the cfg3 bench (3 steps) -> gpurun_out/<tag>/kernel_stats_cfg3.md
tag=${1:-r05p}; cfg=${2:-cfg3}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
here=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $here/$out/prof -o $cfg -- python $here/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-stock-baseline > $here/$out/prof_$cfg.log 2>&1; echo "rocprof rc=$?"
cd $here
db=$(find $out/prof -name "*.db" | head -n 1)
if [ -n "$db" ]; then python tools/rocpd_stats.py $db > $out/kernel_stats_$cfg.md 2>> $out/prof_$cfg.log; fi
head -n 60 $out/kernel_stats_$cfg.md | cut -c1-190
find $out/prof -size +20M -delete
