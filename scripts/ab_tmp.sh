cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s3
R=$GRAFT_REPO_ROOT/safemotionsrisk_amd
for c in reach reach_brake reach_risk; do
  echo "config=$c"
  python scripts/bench_env.py SMR_LIB $R/libsmr_hip.so,$R/libsmr_t1.so,$R/libsmr_hip.so,$R/libsmr_t1.so $c --steps 200
done
