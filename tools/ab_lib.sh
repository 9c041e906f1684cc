#!/bin/bash
# A/B of two builds of libmi355spmv.so: navierstokes_amd/csrc/ab_old/libmi355spmv.so (built from an earlier commit in a worktree) against
# the tree's own, alternating processes, several rounds.  usage: bash tools/ab_lib.sh "<n> <kind>" ["<n> <kind>" ...]
cd "${GRAFT_REPO_ROOT:-.}"
OLD=$PWD/navierstokes_amd/csrc/ab_old/libmi355spmv.so; NEW=$PWD/navierstokes_amd/csrc/libmi355spmv.so
for cfg in "$@"; do
  for round in $(seq 1 ${MI355_AB_ROUNDS:-3}); do
    MI355_LIB_TAG=old MI355_SPMV_LIBRARY=$OLD timeout -k 10 120 python tools/ab_lib.py $cfg 2>&1 | grep ABLIB || exit 1
    MI355_LIB_TAG=new MI355_SPMV_LIBRARY=$NEW timeout -k 10 120 python tools/ab_lib.py $cfg 2>&1 | grep ABLIB || exit 1
  done
done
