#!/bin/bash
# Round 6 gate: the WHOLE GPU suite in the driver's order (-x, as the driver runs it), smoke, then the driver's bench command without the CPU leg.
#   gpurun --timeout 1500 -- 'bash tools/r06_gate.sh r06b'
tag=${1:-r06gate}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1000 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -n 4 $out/pytest_gpu.log
timeout 120 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.log
timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_cmd_no_cpu_leg.json 2> $out/bench_driver_cmd.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("$out/bench_driver_cmd_no_cpu_leg.json").read().strip().splitlines()[-1])
    print("cfg3 ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"])
    for k, v in d.get("other_configs", {}).items():
        print(k, v.get("ms_per_step"), v.get("roofline_frac"))
except Exception as e:
    print("bench parse failed", e)
PY
