"""relation_metrics on pose0-syn-shaped scores (the train list: E = 1,996,020 in 964 type-sorted blocks; the test list a ninth of it):
wall time per call, for `tools/prof.sh stats` to list the launches behind it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gripnet_amd.synth import add_pose_test_split, make_pose
from gripnet_amd.utils import relation_metrics

dev = torch.device("cuda:0")
data = add_pose_test_split(make_pose(sys.argv[1] if len(sys.argv) > 1 else "pose0-syn"))
gen = torch.Generator().manual_seed(3)
for name, rl in (("train", data.train_range), ("test", data.test_range)):
    E = int(rl[-1, 1])
    pos = torch.sigmoid(torch.randn(E, generator=gen) + 0.3).to(dev)
    neg = torch.sigmoid(torch.randn(E, generator=gen)).to(dev)
    pos[: E // 50] = pos[0]                                   # ties
    for _ in range(3):
        out = relation_metrics(pos, neg, rl)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        out = relation_metrics(pos, neg, rl)
    torch.cuda.synchronize()
    print(name, "E =", E, "us per call", round(1e5 * (time.perf_counter() - t), 1), "mean auroc", float(out[1].nanmean()))
