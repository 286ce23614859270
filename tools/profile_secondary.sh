root=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_nc gpurun_out/prof_tr gpurun_out/prof_fb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nc -- python3 tools/bench_nc.py --iters 20 > gpurun_out/prof_nc.log 2>&1 && echo ok1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fb -- python3 tools/bench_nc.py --model freebase-c --iters 20 > gpurun_out/prof_fb.log 2>&1 && echo ok2 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tr -- python3 tools/bench_train.py --steps 20 > gpurun_out/prof_tr.log 2>&1 && echo ok3
for t in nc fb tr; do python3 tools/summarize_prof.py gpurun_out/prof_$t > gpurun_out/prof_${t}_stats.md; done
tail -3 gpurun_out/prof_nc.log gpurun_out/prof_fb.log gpurun_out/prof_tr.log
