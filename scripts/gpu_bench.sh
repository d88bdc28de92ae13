#!/bin/bash
# bench lines only (GPU box).  usage: BENCH_CONFIGS="space ball" bash scripts/gpu_bench.sh [tag]
cd "$GRAFT_REPO_ROOT"
T=${1:-run}; O=gpurun_out/bench_$T; mkdir -p $O
for c in ${BENCH_CONFIGS:-space space_bm ball ball_bm reach}; do
  timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline > $O/$c.json 2> $O/$c.err || echo "bench $c failed"
  python -c "import json,sys; d=json.load(open('$O/$c.json')); print('$T $c', round(d['value']/1e6,2), 'M env-steps/s', {k: round(v,3) for k,v in d['kernel_ms'].items() if v})" || tail -3 $O/$c.err
done
