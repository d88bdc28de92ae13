#!/bin/bash
# A/B of the jobs kernel's prepared-run pool (GPU box): parity suite under SMR_JOB_POOL=1, then bench lines old / pool at several refill thresholds
# usage: ab_pool.sh [LIB] ; output gpurun_out/r4/ab_pool.log
LIBV=${1:-$GRAFT_REPO_ROOT/build_variants/pool/libsmr_hip.so}
O=gpurun_out/r4; mkdir -p $O
export SMR_LIB=$LIBV
SMR_JOB_POOL=1 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py -x -q -m gpu > $O/ab_pool_tests.log 2>&1 || { tail -30 $O/ab_pool_tests.log; exit 1; }
tail -3 $O/ab_pool_tests.log
: > $O/ab_pool.log
for cfg in space space_n1 reach; do
  SMR_JOB_POOL=0 python scripts/bench_env.py SMR_JOB_POOL 0 $cfg >> $O/ab_pool.log 2>&1
  for r in 1 16 32 64; do
    SMR_JOB_POOL=1 python scripts/bench_env.py SMR_JOB_REFILL_IDLE $r $cfg >> $O/ab_pool.log 2>&1
  done
done
SMR_JOB_POOL=0 python scripts/bench_env.py SMR_JOB_POOL 0 space4096 >> $O/ab_pool.log 2>&1
SMR_JOB_POOL=1 python scripts/bench_env.py SMR_JOB_REFILL_IDLE 1,16,64 space4096 >> $O/ab_pool.log 2>&1
cat $O/ab_pool.log
