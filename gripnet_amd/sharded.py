"""Relation-sharded forward of the PoSE model across the GPUs of one node.

The reference is single-device (no torch.distributed anywhere); this is the multi-GPU form of the
same forward.  The dd aggregation of myRGCN is a sum over edges (gripnet/layers.py:131,178-189),
linear in edge subsets, so:

  * the type-sorted dd edge list is cut into ``world_size`` contiguous edge ranges balanced by edge
    count (= relation-id sharding with load balancing for skewed relation sizes);
  * every rank runs the gene layers (gg, gd) itself: 45 of the step's ~110 us on pose0-syn; sharding them by
    destination rows would need two 1.2 MB all-gathers per forward, which cost more than they save at this size;
  * every rank computes the UN-normalised partial ``P_k[n_d, out]`` of its edge range;
  * ONE ``all_reduce(SUM)`` of ``P_k`` (82,560 B at n_d = 645: latency-bound over xGMI);
  * every rank finalises (global mean, root, bias, ReLU; the in-degree is over the full graph) and
    scores its own edge range with the DistMult decoder: no exchange for the scores.

One process per GPU, launched by ``torch.distributed.run``; backend "nccl" (= RCCL on ROCm) on the
GPU box, "gloo" in the CPU tests.  The arithmetic is behind ``kernels`` so that the orchestration
(ranges, collective, slot layout) is testable without a GPU; the default is the HIP path.
"""
from __future__ import annotations

from typing import Optional

import torch

from .utils import shard_edge_ranges


class _ShardWeights:
    """Hands the shard's relational plan to the external layer's launch, which computes W_r = att . basis on the side
    (gn_graph_aggregate_with_rgcn_weights_f32).  encode_genes always runs before partial, also when it is replayed
    as a hipGraph, so once the combined launch has been used the weights are known to be in the workspace."""

    def __init__(self, plan, conv):
        self.plan, self.conv, self.ready = plan, conv, False

    def cowork_request(self):
        return (self.plan, self.conv.basis, self.conv.att)

    def cowork_done(self):
        self.ready = True


class HipShardKernels:
    """The product arithmetic: HIP kernels behind the C ABI (no CPU fallback)."""

    def __init__(self, model, data, lo, hi):
        from . import _hip
        self._hip, self.model, self.data = _hip, model, data
        self.conv = model.dd.conv_list[0]
        self.plan = _hip.RgcnPlan(data.train_idx, data.train_range, data.n_d_node, lo, hi)
        self.weights = _ShardWeights(self.plan, self.conv)
        self.idx = data.train_idx[:, lo:hi].contiguous()
        self.et = data.train_et[lo:hi].contiguous()

    def encode_genes(self):
        z = self.model.gg(None, self.data.gg_edge_index, edge_weight=self.data.edge_weight, if_catout=True)
        return self.model.gd(z, self.data.gd_edge_index, mod="cat", if_relu=True, _cowork=self.weights)

    def partial(self, x, out):
        c = self.conv
        return self.plan.forward(x, c.basis, c.att, None, None, False, out, partial=True,
                                 weights_ready=self.weights.ready)

    def finalize(self, summed, x, out, slot0):
        c = self.conv                                           # concat slot 0 is copied by the same launch
        return self.plan.finalize(summed, x, c.root, c.bias, True, out, side=(x, slot0, 0))

    def score(self, z, sigmoid=True):
        return self.model.dmt(z, self.idx, self.et, sigmoid=sigmoid)


class ShardedPoseForward:
    """``z, score_of_my_edge_range = fwd()`` on every rank; ``z`` is identical on all ranks."""

    def __init__(self, model, data, rank: int, world_size: int, group=None, kernels=None):
        self.rank, self.world_size, self.group = int(rank), int(world_size), group
        self.n_d = int(data.n_d_node)
        E = int(data.train_idx.shape[1])
        self.edge_lo, self.edge_hi = shard_edge_ranges(E, self.world_size)[self.rank]
        self.kernels = kernels if kernels is not None else HipShardKernels(model, data, self.edge_lo, self.edge_hi)
        conv = model.dd.conv_list[0]
        self.in_dim, self.out_dim = conv.in_channels, conv.out_channels
        dev = data.train_idx.device
        self._partial = torch.empty((self.n_d, self.out_dim), dtype=torch.float32, device=dev)

    def all_reduce(self, t: torch.Tensor):
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def __call__(self, sigmoid: bool = True):
        k = self.kernels
        x = k.encode_genes()                                          # [n_d, in_dim], replicated
        out = torch.empty((self.n_d, self.in_dim + self.out_dim), dtype=torch.float32, device=x.device)
        k.partial(x, self._partial)                                   # un-normalised sum over my edge range
        self.all_reduce(self._partial)                                # the one exchange step of the path
        # mean / root / bias / ReLU (layers.py:191-197,305) and concat slot 0 (layers.py:264-266)
        k.finalize(self._partial, x, out[:, self.in_dim:], out[:, :self.in_dim])
        return out, k.score(out, sigmoid=sigmoid)
