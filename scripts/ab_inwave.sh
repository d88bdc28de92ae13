#!/bin/bash
# in-wave cooperative finishing of long runs (SMR_JOB_INWAVE=T): parity suite under it, then bench lines per batch size (GPU box)
O=gpurun_out/r4; mkdir -p $O
SMR_JOB_INWAVE=${INWAVE_TEST:-4} timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py tests/test_gpu_policy.py -x -q -m gpu > $O/ab_inwave_tests.log 2>&1 || { tail -30 $O/ab_inwave_tests.log; exit 1; }
tail -2 $O/ab_inwave_tests.log
: > $O/ab_inwave.log
for b in 4096 16384 32768 65536; do
  echo "== space, $b envs" >> $O/ab_inwave.log
  python scripts/bench_env.py SMR_JOB_INWAVE 0,3,4,6 space --envs-per-gpu $b >> $O/ab_inwave.log 2>&1
done
echo "== reach" >> $O/ab_inwave.log; python scripts/bench_env.py SMR_JOB_INWAVE 0,4 reach >> $O/ab_inwave.log 2>&1
echo "== space_bm" >> $O/ab_inwave.log; python scripts/bench_env.py SMR_JOB_INWAVE 0,4 space_bm >> $O/ab_inwave.log 2>&1
cut -c1-60,150-330 $O/ab_inwave.log
