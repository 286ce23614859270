#!/usr/bin/env python3
"""Run the seeded property tests of tests/test_gpu_fuzz.py on seeds beyond the suite's (development tool: a bug hunt after a
kernel change).  `python tools/fuzz_sweep.py [first_seed] [count]`"""
import inspect
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as fz                         # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 4
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
gpu = torch.device("cuda:0")
failed = 0
for name, fn in sorted(vars(fz).items()):
    if not name.startswith("test_fuzz") or not callable(fn):
        continue
    params = inspect.signature(fn).parameters
    if "seed" not in params:
        continue
    for seed in range(first, first + count):
        t0 = time.perf_counter()
        try:
            fn(gpu, seed) if list(params)[:2] == ["gpu", "seed"] else fn(gpu=gpu, seed=seed)
            print("{} seed {}: ok ({:.1f} s)".format(name, seed, time.perf_counter() - t0), flush=True)
        except AssertionError as err:
            failed += 1
            print("{} seed {}: FAILED {}".format(name, seed, err), flush=True)
sys.exit(1 if failed else 0)
