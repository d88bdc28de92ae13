#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/exp; mkdir -p $O
for c in space space_n1; do for sync in 0 1; do for envs in 65536 262144; do
  SMR_SYNC_RESET=$sync python bench.py --config $c --envs-per-gpu $envs --steps 100 --warmup 20 --no-cpu-baseline > $O/x.json 2>$O/x.err
  python -c "import json; d=json.load(open('$O/x.json')); print('$c sync=$sync envs=$envs', round(d['value']/1e6,2), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done; done; done
