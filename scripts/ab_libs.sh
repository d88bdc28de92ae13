#!/bin/bash
# A/B of two builds of the library on one box: full GPU suite on the tree's build, then bench lines base / new per configuration
# usage: ab_libs.sh "CONFIG ..." [BASE_LIB] ; output gpurun_out/r4/ab_libs.log
O=gpurun_out/r4; mkdir -p $O
BASE=${2:-$GRAFT_REPO_ROOT/build_variants/base/libsmr_hip.so}; NEW=$GRAFT_REPO_ROOT/safemotionsrisk_amd/libsmr_hip.so
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/ab_libs_tests.log 2>&1 || { tail -40 $O/ab_libs_tests.log; exit 1; }
tail -3 $O/ab_libs_tests.log
fi
: > $O/ab_libs.log
for cfg in ${1:-space space_n1 reach space_bm ball ball_bm reach_brake space4096}; do
  python scripts/bench_env.py SMR_LIB $BASE,$NEW $cfg >> $O/ab_libs.log 2>&1
done
sed -e 's#SMR_LIB=[^ ]*/build_variants/base/libsmr_hip.so#BASE#' -e 's#SMR_LIB=[^ ]*/safemotionsrisk_amd/libsmr_hip.so#NEW #' $O/ab_libs.log | cut -c1-330
