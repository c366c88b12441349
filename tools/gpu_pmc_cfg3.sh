#!/bin/bash
# PMC passes of the cfg3 step only (rocprofv3 --kernel-trace --pmc, one counter set per pass): the last block of tools/gpu_evidence.sh.
#   gpurun --timeout 400 -- 'bash tools/gpu_pmc_cfg3.sh <tag>'
tag=${1:-pmc}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
here=$PWD
cd /tmp
IFS=';' read -ra sets <<< "FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
i=0
for ctrs in "${sets[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs -d $here/$out/pmc$i -o cfg3 -- python $here/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-stock-baseline > $here/$out/pmc$i.log 2>&1; echo "pmc [$ctrs] rc=$?"
  db=$(find $here/$out/pmc$i -name "*.db" | head -n 1)
  [ -n "$db" ] && python $here/tools/rocpd_pmc.py $db > $here/$out/pmc$i.md 2>> $here/$out/pmc$i.log
  head -n 6 $here/$out/pmc$i.md | cut -c1-160
  find $here/$out/pmc$i -size +20M -delete
done
