#!/bin/bash
# dev helper: the relational kernel's rocprofv3 time for the product library and every variant, twice each, on one box
for rep in 1 2; do
for lib in gripnet_amd/lib/libgripnet_hip.so gripnet_amd/lib/libgripnet_hip_*.so; do
  case $lib in *stamps*) continue;; esac
  echo "== $lib"
  GN_HIP_LIBRARY=$PWD/$lib TOPN=30 tools/prof_stats.sh ab_$(basename $lib .so) tools/bench_kernels.py --what rgcn --iters 40 | grep -e k_rgcn
done; done
