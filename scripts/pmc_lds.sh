#!/bin/bash
# LDS counters of the step kernels (GPU box): is the jobs kernel bound by the LDS pipe?  usage: pmc_lds.sh [CONFIG] [TAG]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C=${1:-space}; T=${2:-base}
O=gpurun_out/r4/pmc_lds_${C}_$T; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O -- python3 bench.py --config $C --steps 40 --warmup 5 --no-cpu-baseline > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "smr::" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k, {c: round(acc[k][c] / n[k][c]) for c in sorted(acc[k])})
PY
