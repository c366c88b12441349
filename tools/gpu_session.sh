#!/bin/bash
# A short GPU-box session: selected tests (pytest -k expression / files in $TESTS), then quick bench lines.
#   gpurun --timeout 900 -- 'TESTS="tests/test_gpu_loop.py" CONFIGS="cfg5a cfg5b" bash tools/gpu_session.sh <tag>'
#   env: TESTS (pytest args, default none; or TESTS_FILE = a file holding them, for -k expressions with quotes), CONFIGS (bench configs, default none), STEPS (default 6), BENCH_ARGS, EXTRA (command run last)
tag=${1:-s}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ -n "$TESTS_FILE" ]; then TESTS=$(cat $TESTS_FILE); fi
if [ -n "$TESTS" ]; then
  eval timeout ${TEST_TIMEOUT:-600} python -m pytest $TESTS -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
  grep -E "passed|failed|FAILED|Error" $out/pytest.log | tail -n 45
fi
for cfg in $CONFIGS; do
  timeout 300 python bench.py --config $cfg --steps ${STEPS:-6} --warmup 3 --no-cpu-baseline --no-stock-baseline $BENCH_ARGS > $out/bench_$cfg.json 2> $out/bench_$cfg.err; echo "bench $cfg rc=$?"
  python - <<PY
import json
try:
    j = json.loads([l for l in open("$out/bench_$cfg.json") if l.startswith("{")][-1])
    r = j.get("roofline", {})
    print("$cfg", j["ms_per_step"], "ms/step median", j.get("ms_per_step_median"), "loss", j.get("ctc_loss_first_step"), "ref", j.get("ctc_loss_ref"), "rel", j.get("ctc_loss_rel_diff"))
    for k, v in (r.get("recurrent_kernels") or {}).items():
        print("   ", k, v)
except Exception as e:
    print("$cfg: no line:", e); print(open("$out/bench_$cfg.err").read()[-1500:])
PY
done
if [ -n "$EXTRA" ]; then bash -c "$EXTRA" > $out/extra.log 2>&1; echo "extra rc=$?"; tail -n 60 $out/extra.log; fi
