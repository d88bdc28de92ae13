#!/bin/bash
# how much the staged refill costs the step kernels, and whether one refill block per CU helps (GPU box)
O=gpurun_out/r4; mkdir -p $O; : > $O/ab_side.log
for cfg in ${1:-space reach space_bm ball_bm}; do
  echo "== $cfg: SMR_SYNC_RESET (1 = the refill on the critical path, step kernels alone)" >> $O/ab_side.log
  python scripts/bench_env.py SMR_SYNC_RESET 0,1 $cfg >> $O/ab_side.log 2>&1
  echo "== $cfg: SMR_SIDE_LDS_PAD" >> $O/ab_side.log
  python scripts/bench_env.py SMR_SIDE_LDS_PAD 0,24576,36864 $cfg >> $O/ab_side.log 2>&1
done
cut -c1-330 $O/ab_side.log
