#!/bin/bash
# One GPU-box session: parity tests, then the bench configurations.  Usage (from the repo root, through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag> [pytest-args...]'
tag=${1:-run}; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s --maxfail=25 --tb=short "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -n 40 $out/pytest.log
for cfg in cfg3 cfg2 cfg5a cfg5b; do
  steps=20; [ $cfg != cfg3 ] && steps=6
  timeout 300 python bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err; echo "bench $cfg rc=$?"
  tail -c 1500 $out/bench_$cfg.json; echo
done
