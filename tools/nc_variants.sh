#!/bin/bash
# dev helper: NC forward times for the product library and every experimental one (make VARIANT=name VFLAGS=...)
for lib in gripnet_amd/lib/libgripnet_hip.so gripnet_amd/lib/libgripnet_hip_*.so; do
  case $lib in *stamps*) continue;; esac
  echo "== $lib"
  GN_HIP_LIBRARY=$PWD/$lib python3 tools/bench_nc.py 2>/dev/null
  GN_HIP_LIBRARY=$PWD/$lib python3 tools/bench_nc.py --model freebase-c 2>/dev/null
done
