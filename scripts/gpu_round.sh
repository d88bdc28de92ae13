#!/bin/bash
# one GPU-box call: the GPU test suite, then bench lines of the main configurations.  Output under gpurun_out/$ROUND_DIR/ (default r3).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${ROUND_DIR:-r3}; mkdir -p $O
timeout -k 10 ${TEST_TIMEOUT:-900} python -u -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log
grep -E 'passed|failed|error' $O/gpu_tests.log | tail -5
for c in ${BENCH_CONFIGS:-space space_n3 space_n1}; do
  timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err || echo "bench $c failed"
  python -c "import json,sys; d=json.load(open('$O/bench_$c.json')); print('$c', round(d['value']/1e6,2), 'M env-steps/s', {k: round(v,4) for k,v in d['kernel_ms'].items() if v})" || tail -3 $O/bench_$c.err
done
