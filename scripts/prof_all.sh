#!/bin/bash
# profile sets of several configurations in one GPU-box call: bash scripts/prof_all.sh "space space_n1 ..." (see gpu_profile.sh)
cd "$GRAFT_REPO_ROOT"
for c in $1; do
  case $c in space4096) E=4096; N=24;; space_n1) E=65536; N=1;; space_n3) E=65536; N=3;; *) E=65536; N=24;; esac
  echo "=== $c"; bash scripts/gpu_profile.sh $c $E $N 2>&1 | tail -4
done
