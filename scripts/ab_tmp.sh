cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s3
for c in ball_bm space_bm reach; do
  echo "config=$c"
  python scripts/bench_env.py SMR_RESET_GROUP_STRIDE 1,2,4 $c --steps 200
done
