#!/bin/bash
# PMC passes over tools/bench_conv2_wgrad.py (the three conv2 kernels alone): where do a wave's cycles go?
#   gpurun --timeout 400 -- 'bash tools/gpu_pmc_conv.sh <tag>'
tag=${1:-pmc_conv}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
here=$PWD
cd /tmp
IFS=';' read -ra sets <<< "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY;SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS;FETCH_SIZE;SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"
i=0
for ctrs in "${sets[@]}"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $ctrs -d $here/$out/pmc$i -o conv -- python $here/tools/bench_conv2_wgrad.py > $here/$out/pmc$i.log 2>&1; echo "pmc [$ctrs] rc=$?"
  db=$(find $here/$out/pmc$i -name "*.db" | head -n 1)
  [ -n "$db" ] && python $here/tools/rocpd_pmc.py $db conv > $here/$out/pmc$i.md 2>> $here/$out/pmc$i.log
  grep -v "colsum\|elementwise" $here/$out/pmc$i.md | head -n 30 | cut -c1-170
  find $here/$out/pmc$i -size +20M -delete
done
