cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s3
R=$GRAFT_REPO_ROOT/safemotionsrisk_amd
for c in ball_bm ball; do
  echo "config=$c"
  python scripts/bench_env.py SMR_LIB $R/libsmr_hip.so,$R/libsmr_A.so,$R/libsmr_B.so,$R/libsmr_hip.so $c --steps 200
done
