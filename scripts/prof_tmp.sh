cd "$GRAFT_REPO_ROOT"
export SMR_LIB=$PWD/build_variants/libsmr_prof.so
for c in space space_bm ball_bm; do echo "== $c"; python scripts/profile_rows.py $c 30 2>&1 | grep -E "reset:|target point|staged|spawn|config"; done
