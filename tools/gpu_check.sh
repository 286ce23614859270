#!/bin/bash
# bring-up helper: GPU parity tests + default bench; logs under gpurun_out/
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest.log 2>&1; echo pytest=$?; tail -${TAILN:-6} gpurun_out/pytest.log
timeout 600 python bench.py ${BENCH_ARGS:---no-cpu-baseline} > gpurun_out/bench.log 2>&1; echo bench=$?
grep '^{' gpurun_out/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('value','ms_per_step','roofline','entry_point_us_per_step','cpu_baseline','parity'): print(k, d.get(k))" || tail -20 gpurun_out/bench.log
