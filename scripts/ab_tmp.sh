cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s3
R=$GRAFT_REPO_ROOT/safemotionsrisk_amd
for c in space space_bm reach space_n1; do
  echo "config=$c"
  SMR_LIB=$R/libsmr_q4.so SMR_JOB_BLOCK_LANES=512 python scripts/bench_env.py SMR_JOB_BLOCKS 256,512,384,256,512 $c --steps 200
done
