#!/bin/bash
# the round's profile set (GPU box): kernel stats, three PMC passes, the bench line.  Output under gpurun_out/final/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/stats.log 2>&1
for g in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  d=$O/pmc_$(echo $g | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $g -d $d -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $d.log 2>&1
done
python3 scripts/make_pmc_json.py $O/pmc.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU
cp $O/pmc.json profiles/round1_final_pmc.json        # bench.py reads the traffic figure from here
python3 bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
