#!/bin/bash
# dev helper: relational-layer parity + kernel time for the product library and every experimental one
#   (make VARIANT=name VFLAGS=...).  usage: tools/variants.sh [what]   (what: bench_kernels --what list, default rgcn)
for lib in gripnet_amd/lib/libgripnet_hip.so gripnet_amd/lib/libgripnet_hip_*.so; do
  [ -f $lib ] || continue
  echo "== $lib"
  GN_HIP_LIBRARY=$PWD/$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "rgcn or kernel_paths or pose0" 2>&1 | tail -1
  GN_HIP_LIBRARY=$PWD/$lib TOPN=6 tools/prof_stats.sh v_$(basename $lib .so) tools/bench_kernels.py --what ${1:-rgcn} --iters 20 | grep -e k_rgcn -e distmult_lds -e aggregate
done
