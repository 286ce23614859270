for v in "" cma cmb cmc cmd; do
  if [ -n "$v" ]; then export GN_HIP_LIBRARY=$PWD/gripnet_amd/lib/libgripnet_hip_$v.so; else unset GN_HIP_LIBRARY; fi
  echo "variant ${v:-default}"
  tools/prof_stats.sh cm$v bench.py --steps 30 --warmup 3 --no-cpu-baseline --launch eager 2>&1 | grep -E "k_rgcn_acc<48, 2, true>"
done
