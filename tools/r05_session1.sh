#!/bin/bash
# round 5, session 1: the sparse-instruction probe, the 8-clip sweeps in both forms (tests + timings), a short cfg3 bench
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python tools/probe_smfmac.py > $out/smfmac.txt 2>&1; echo "probe rc=$?"; head -n 5 $out/smfmac.txt; tail -n 4 $out/smfmac.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "persistent_sweeps or sparse_and_dense or persistent_initial_state" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 15 $out/pytest.log
timeout 300 python tools/time_sweeps.py gru,2,32,1024,751 variant=32 gru,2,32,1024,751 variant=0 gru,2,32,1024,751,ragged variant=32 gru,2,32,1024,751,ragged variant=0 lstm,2,32,1024,401 variant=32 lstm,2,32,1024,401 > $out/time_sweeps.txt 2>&1; echo "time_sweeps rc=$?"; cat $out/time_sweeps.txt
timeout 300 python bench.py --config cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-stock-baseline > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "bench rc=$?"; tail -c 1500 $out/bench_cfg3.json
