#!/bin/bash
# Round 6: the general sweeps after the AGPR pin + structured-sparse sets: kernel tests, kernel-level timing (sparse vs dense routing), bench legs.
tag=${1:-r06c}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "persist3 or persistent or sparse" > $out/pytest_persist.log 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest_persist.log
timeout 300 python tools/time_sweeps.py lstm,1,64,1280,751,ragged lstm,2,64,1280,751,ragged gru,2,64,1024,751,ragged gru,2,32,800,201 variant=64 lstm,1,64,1280,751,ragged gru,2,64,1024,751,ragged > $out/time_sweeps.txt 2>&1; cat $out/time_sweeps.txt
for cfg in cfg5b cfg5a; do
  timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-stock-baseline --no-other-configs > $out/bench_$cfg.json 2> $out/bench_$cfg.err; echo "bench $cfg rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("$out/bench_$cfg.json").read().strip().splitlines()[-1])
    print("$cfg ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"].get("kernel"), d["roofline"].get("us_per_launch"))
except Exception as e:
    print("bench parse failed", e)
PY
done
