#!/bin/bash
# kernel-time split of the deferred-run path (GPU box): rocprofv3 kernel stats of bench.py under SMR_JOB_DEFER
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
for cfg in ${CFGS:-space reach}; do
for d in ${DEFERS:-0 1 3}; do
  export SMR_JOB_DEFER=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_defer_${cfg}_$d -- python3 bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline > $O/stats_defer_${cfg}_$d.log 2>&1
  f=$(find $O/stats_defer_${cfg}_$d -name "*kernel_stats.csv" | head -1)
  echo "== $cfg defer=$d"; head -8 "$f" | cut -c1-160
done
done
