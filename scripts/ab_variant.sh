#!/bin/bash
# A/B of the tree's build against a variant build (GPU box): usage ab_variant.sh VARIANT_DIR "CONFIG ..."
O=gpurun_out/r4; mkdir -p $O
V=$GRAFT_REPO_ROOT/build_variants/$1/libsmr_hip.so; NEW=$GRAFT_REPO_ROOT/safemotionsrisk_amd/libsmr_hip.so
: > $O/ab_variant_$1.log
for cfg in ${2:-space space_bm reach}; do
  python scripts/bench_env.py SMR_LIB $NEW,$V $cfg >> $O/ab_variant_$1.log 2>&1
done
sed -e "s#SMR_LIB=[^ ]*/build_variants/$1/libsmr_hip.so#VAR #" -e 's#SMR_LIB=[^ ]*/safemotionsrisk_amd/libsmr_hip.so#TREE#' $O/ab_variant_$1.log | cut -c1-330
