#!/usr/bin/env python3
"""dev helper: decoder gradients at pose2-syn scale (8.4 M edges: the unstaged scatter) against torch index_add_ on the GPU."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gripnet_amd import _hip
from gripnet_amd.synth import make_pose
dev = torch.device("cuda:0")
for wl in ("pose0-syn", "pose2-syn"):
    data = make_pose(wl).to(dev)
    n, R, f = data.n_d_node, data.n_dd_edge_type, 80
    torch.manual_seed(3)
    z = torch.randn(n, f, device=dev) * 0.3; D = torch.randn(R, f, device=dev) * 0.3
    ei, et = data.train_idx, data.train_et
    g = torch.randn(ei.shape[1], device=dev)
    dz = torch.empty_like(z); dd = torch.empty_like(D)
    _hip.distmult_backward(z, ei, et, D, g, dz, dd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        _hip.distmult_backward(z, ei, et, D, g, dz, dd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    u, v = ei[0], ei[1]
    rz = torch.zeros(n, f, device=dev, dtype=torch.float64); rd = torch.zeros(R, f, device=dev, dtype=torch.float64)
    zz, DD, gg = z.double(), D.double(), g.double()
    step = 1 << 20
    for a in range(0, ei.shape[1], step):
        s = slice(a, a + step)
        rz.index_add_(0, u[s], gg[s, None] * zz[v[s]] * DD[et[s]]); rz.index_add_(0, v[s], gg[s, None] * zz[u[s]] * DD[et[s]])
        rd.index_add_(0, et[s], gg[s, None] * zz[u[s]] * zz[v[s]])
    print(wl, "E", ei.shape[1], "dz rel err {:.2e}".format(((dz.double() - rz).abs().max() / rz.abs().max()).item()),
          "dD rel err {:.2e}".format(((dd.double() - rd).abs().max() / rd.abs().max()).item()), "{:.0f} us per call".format(dt * 1e6))
