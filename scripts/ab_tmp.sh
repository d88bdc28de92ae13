cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q > gpurun_out/ab/tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/ab/tests.log
BENCH_CONFIGS="space_bm ball_bm reach reach_risk" bash scripts/gpu_bench.sh r3ag
