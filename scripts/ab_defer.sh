#!/bin/bash
# A/B of the deferred runs (GPU box): parity suite under SMR_JOB_DEFER, then bench lines at several thresholds
LIBV=${1:-$GRAFT_REPO_ROOT/build_variants/defer/libsmr_hip.so}
O=gpurun_out/r4; mkdir -p $O
export SMR_LIB=$LIBV
SMR_JOB_DEFER=${DEFER_TEST:-3} timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_harness.py tests/test_gpu_policy.py -x -q -m gpu > $O/ab_defer_tests.log 2>&1 || { tail -30 $O/ab_defer_tests.log; exit 1; }
tail -3 $O/ab_defer_tests.log
: > $O/ab_defer.log
for cfg in ${CFGS:-space space_n1 reach space_bm space4096}; do
  python scripts/bench_env.py SMR_JOB_DEFER ${DEFERS:-0,2,3,4,6} $cfg >> $O/ab_defer.log 2>&1
done
cat $O/ab_defer.log | cut -c1-300
