#!/bin/bash
# PMC passes (separate runs, kernel-trace only: MI355X_MICROARCH.md rocprofv3 section) of one bench configuration, filtered to kernels
# whose name contains $FILTER.   gpurun -- 'CONFIG=cfg5a FILTER=persist3 SETS="A B C;D E" bash tools/gpu_pmc.sh <tag>'
tag=${1:-pmc}
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
here=$PWD; cd /tmp
IFS=';' read -ra sets <<< "${SETS:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE}"
i=0
for ctrs in "${sets[@]}"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $ctrs -d $out/p$i -o run -- python $here/bench.py --config ${CONFIG:-cfg3} --steps 1 --warmup 1 --no-cpu-baseline --no-stock-baseline > $out/p$i.log 2>&1; echo "pmc [$ctrs] rc=$?"
  db=$(find $out/p$i -name "*.db" | head -n 1)
  [ -n "$db" ] && python $here/tools/rocpd_pmc.py $db ${FILTER:-} > $out/p$i.md 2>> $out/p$i.log
  cat $out/p$i.md | cut -c1-200
  find $out/p$i -size +20M -delete
done
