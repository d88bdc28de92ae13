#!/bin/bash
# jobs kernel: blocks per CU x jobs per hand-out (GPU box)
O=gpurun_out/r4; mkdir -p $O
: > $O/sweep_quota.log
for cfg in ${CFGS:-space reach}; do
for blocks in 256 512; do
  echo "== $cfg SMR_JOB_BLOCKS=$blocks" >> $O/sweep_quota.log
  SMR_JOB_BLOCKS=$blocks python scripts/bench_env.py SMR_JOB_QUOTA_MAX 64,48,32,16 $cfg >> $O/sweep_quota.log 2>&1
done
done
cut -c1-60,150-330 $O/sweep_quota.log
