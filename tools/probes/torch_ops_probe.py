"""Development probe: which aten ops torch itself still launches inside an eager PoSE training step (torch profiler, one step):
the autograd engine's sums of two gradients of one tensor, copies made contiguous.
    python tools/probes/torch_ops_probe.py
"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from gripnet_amd import _hip
from gripnet_amd.pipeline import PoseModel
from gripnet_amd.synth import make_pose
from gripnet_amd.utils import link_loss
from gripnet_amd.optim import Adam
dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
opt = Adam(model.parameters(), lr=0.01)
sampler = _hip.NegativeSampler(data.train_idx, data.n_d_node, data.train_range)
neg = sampler.sample(seed=0)
drawn = torch.ones((1,), dtype=torch.int64, device=dev)
one = torch.ones((), dtype=torch.float32, device=dev)
def step():
    sampler.sample(seed=0, out=neg, step=drawn)
    opt.zero_grad()
    z = model.encode(data)
    pos = model.dmt(z, data.train_idx, data.train_et)
    negs = model.dmt(z, neg, data.train_et)
    loss = link_loss(pos, negs)
    loss.backward(one)
    opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = prof.events()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::"):
        ks = [k.name[:60] for k in e.kernels] if hasattr(e, "kernels") else []
        print(e.name, e.input_shapes, "parent:", e.cpu_parent.name[:60], "| stack:", [s for s in (e.stack or [])[:3]])
