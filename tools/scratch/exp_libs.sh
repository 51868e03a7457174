#!/bin/bash
# time the Stage K kernels with experimental builds of the library swapped in (exp_libs/libexp*.so)
R=$GRAFT_REPO_ROOT
cp $R/chomp_amd/libchomp_mi355x.so /tmp/orig.so
for e in "$@"; do
  cp $R/exp_libs/libexp$e.so $R/chomp_amd/libchomp_mi355x.so
  touch $R/chomp_amd/libchomp_mi355x.so
  echo "== exp $e"
  bash $R/tools/scratch/kstats.sh exp$e | grep -E "k_nu_table|k_epoch_init|k_sigma_nodes|value"
done
cp /tmp/orig.so $R/chomp_amd/libchomp_mi355x.so
