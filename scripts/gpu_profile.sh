#!/bin/bash
# profile set of one bench configuration (GPU box): kernel stats + PMC passes.  usage: gpu_profile.sh CONFIG [ENVS CHECKS]
# output under gpurun_out/prof_CONFIG/ (kernel_stats.csv, pmc.json): copy them to profiles/round<R>_kernel_stats_CONFIG.csv / round<R>_pmc_CONFIG.json
C=${1:-space}; ENVS=${2:-65536}; CHECKS=${3:-24}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$C; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config $C --steps 200 --warmup 20 --no-cpu-baseline ${BENCH_EXTRA:-} > $O/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $O/stats.log; exit 1; }
echo "stats done"
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); d=$O/pmc_$i
  rocprofv3 --kernel-trace --output-format csv --pmc $g -d $d -- python3 bench.py --config $C --steps 40 --warmup 5 --no-cpu-baseline ${BENCH_EXTRA:-} > $d.log 2>&1 || { echo "pmc pass $i ($g) failed"; tail -3 $d.log; }
  echo "pmc pass $i done"
done
python3 scripts/make_pmc_json.py $O/pmc.json $C $ENVS $CHECKS $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4          # the raw traces: gpurun copies back at most 64 MiB
head -12 $O/kernel_stats.csv
