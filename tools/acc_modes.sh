#!/bin/bash
# dev helper: time the relational kernel in the product build and in the diagnostic builds (make MODE=1..3)
for m in "" 1 2 3 4 5 6 7 8 16 24; do
  lib=gripnet_amd/lib/libgripnet_hip${m:+_mode$m}.so
  [ -f $lib ] || continue
  echo "== mode ${m:-0}"
  GN_HIP_LIBRARY=$PWD/$lib TOPN=8 tools/prof_stats.sh accm${m:-0} tools/bench_kernels.py --what rgcn --iters 20 | grep -e k_rgcn_acc -e weights
done
