#!/bin/bash
# dev helper: kernel times for the product library and every experimental one (make VARIANT=name VFLAGS=...).
#   usage: tools/variants.sh [what] [pytest -k expression]     (what: bench_kernels --what list, default rgcn)
for lib in gripnet_amd/lib/libgripnet_hip.so gripnet_amd/lib/libgripnet_hip_*.so; do
  [ -f $lib ] || continue
  echo "== $lib"
  [ -n "$2" ] && GN_HIP_LIBRARY=$PWD/$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "$2" 2>&1 | tail -1
  GN_HIP_LIBRARY=$PWD/$lib TOPN=9 tools/prof_stats.sh v_$(basename $lib .so) tools/bench_kernels.py --what ${1:-rgcn} --iters 20 | grep -e k_rgcn -e distmult_lds -e aggregate
done
