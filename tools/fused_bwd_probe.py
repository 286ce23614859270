"""Host time of the pieces of LinkPredictionLossFn.backward (development probe)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gripnet_amd import _hip, autograd
from gripnet_amd.pipeline import PoseModel
from gripnet_amd.synth import make_pose
from gripnet_amd.utils import link_prediction_loss

dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
sampler = _hip.NegativeSampler(data.train_idx, data.n_d_node, data.train_range)
neg = sampler.sample(seed=0)
with torch.no_grad():
    z0 = model.encode(data)
times = {}
orig_call = _hip._call


def timed_call(name, *a, **k):
    t = time.perf_counter()
    r = orig_call(name, *a, **k)
    times[name] = times.get(name, 0.0) + time.perf_counter() - t
    return r


for it in range(6):
    z = z0.clone().requires_grad_(True)
    loss, _, _ = link_prediction_loss(model.dmt, z, data.train_idx, neg, data.train_et)
    torch.cuda.synchronize()
    if it == 5:
        _hip._call = timed_call
    t = time.perf_counter()
    loss.backward()
    t_host = time.perf_counter() - t
    torch.cuda.synchronize()
    print("backward host ms", round(1e3 * t_host, 3), "total ms", round(1e3 * (time.perf_counter() - t), 3))
print({k: round(1e3 * v, 3) for k, v in times.items()})
