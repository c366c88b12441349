#!/bin/bash
# Round 6 mid-round check: kernel + model tests touched by the fp32 coverage / routing changes, then the bench legs of every configuration.
tag=${1:-r06f}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "persist or rnn or sparse" > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -n 3 $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q > $out/pytest_model.log 2>&1; echo "model rc=$?"; tail -n 6 $out/pytest_model.log | cut -c1-300
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stock-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    print("cfg3 ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"])
    for k, v in d.get("other_configs", {}).items():
        print(k, v.get("ms_per_step"), v.get("roofline_frac"))
except Exception as e:
    print("bench parse failed", e)
PY
