#!/bin/bash
# cycle profile of the jobs kernel, old form and prepared-run pool (GPU box; build_variants/poolprof = -DSMR_PROFILE_ROWS build)
O=gpurun_out/r4; mkdir -p $O
export SMR_LIB=$GRAFT_REPO_ROOT/build_variants/poolprof/libsmr_hip.so
for v in "0 64" "1 1" "1 32"; do
  set -- $v
  echo "=== SMR_JOB_POOL=$1 SMR_JOB_REFILL_IDLE=$2" >> $O/prof_pool.log
  SMR_JOB_POOL=$1 SMR_JOB_REFILL_IDLE=$2 timeout -k 10 200 python scripts/profile_rows.py space 50 2>&1 | grep "jobs kernel" >> $O/prof_pool.log
  SMR_JOB_POOL=$1 SMR_JOB_REFILL_IDLE=$2 timeout -k 10 200 python scripts/profile_jobs.py space 2>&1 | grep -v amdgpu.ids | head -8 >> $O/prof_pool.log
done
cat $O/prof_pool.log
