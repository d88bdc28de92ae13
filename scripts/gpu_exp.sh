#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/exp; mkdir -p $O
for g in 32 128 512 1024; do
  SMR_ITEM_GRID=$g python bench.py --config space --steps 200 --warmup 20 --no-cpu-baseline > $O/g$g.json 2>$O/g$g.err
  python -c "import json; d=json.load(open('$O/g$g.json')); print('grid $g', round(d['value']/1e6,2), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done
