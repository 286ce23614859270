#!/usr/bin/env python3
"""The training-step entry of bench.py on its own (development tool; for rocprofv3 traces of the replayed step:
tools/prof.sh train <tag> 20))."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print(json.dumps(bench.train_step_entry(torch.device("cuda:0"), steps)))
