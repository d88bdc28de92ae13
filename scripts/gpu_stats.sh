#!/bin/bash
# kernel-time summary of one bench configuration (GPU box).  usage: gpu_stats.sh CONFIG [TAG] [extra bench args]
C=${1:-space}; T=${2:-$C}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/stats_$T; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/run -- python3 bench.py --config $C --steps 100 --warmup 10 --no-cpu-baseline "$@" > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
find $O/run -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 - <<PY
import csv, json
line = [l for l in open("$O/log.txt") if l.startswith('{"metric"')]
print("== $T:", round(json.loads(line[-1])["value"] / 1e6, 2) if line else "?", "M env-steps/s (under the profiler)")
for r in list(csv.DictReader(open("$O/kernel_stats.csv")))[:8]:
    if "smr::" in r["Name"]:
        print("  %-60s calls %5s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
