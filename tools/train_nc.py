#!/usr/bin/env python3
"""One training step of the node-classification models (the loop body of GripNet-aminer.py:120-147 / GripNet-freebase-c.py:
146-176: forward, class loss, backward, Adam) on aminer-syn, K times (development tool; under tools/prof.sh stats its kernel
table shows which kernels a step launches - the library's own, no library GEMM, no torch index kernels).

    python tools/train_nc.py [aminer|freebase-c] [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gripnet_amd.optim import Adam                                  # noqa: E402
from gripnet_amd.pipeline import AminerModel, FreebaseCModel       # noqa: E402
from gripnet_amd.synth import make_nc                              # noqa: E402
from gripnet_amd.utils import class_loss                           # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "aminer"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
data = make_nc("aminer-syn").to(dev)
torch.manual_seed(1111)
model = (AminerModel(data.n_p_node, data.n_a_node, data.n_a_type) if which == "aminer" else
         FreebaseCModel(data.n_p_node, data.n_q_node, data.n_a_node, data.n_a_type)).to(dev)
opt = Adam(model.parameters(), lr=0.01)
train_nodes = torch.arange(0, data.n_a_node, 2, device=dev)
train_class = data.a_label[train_nodes].contiguous()


def step():
    opt.zero_grad()
    z, score = model(data, train_nodes)
    loss = class_loss(score, train_class)                          # GripNet-aminer.py:133 in one launch each way
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    first = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    last = step()
torch.cuda.synchronize()
print("{} training step: {:.1f} us per step (eager), loss {:.4f} -> {:.4f}".format(which, 1e6 * (time.perf_counter() - t0) / steps,
                                                                                float(first), float(last)))
