"""Decoders with the reference's names and signatures (gripnet/decoder.py), on the gfx950 kernels.

  multiRelaInnerProductDecoder  decoder.py:10-26   DistMult link scorer
  multiClassInnerProductDecoder decoder.py:29-50   linear + softmax node classifier
"""
from __future__ import annotations

import numpy as np
import torch
from torch.nn import Module, Parameter

from . import _hip
from .autograd import ClassLogitsFn, DistMultFn, recording


class multiRelaInnerProductDecoder(Module):
    """``sigmoid(sum_k z[u,k] z[v,k] D[r,k])`` (reference decoder.py:19-23)."""

    def __init__(self, in_dim, num_et):
        super().__init__()
        self.num_et, self.in_dim = num_et, in_dim
        self.weight = Parameter(torch.empty(num_et, in_dim))
        self._seen = []                      # recently scored edge lists: [edge_index, edge_type, versions, plan]
        self.reset_parameters()

    def plan_for(self, z, edge_index, edge_type):
        """Cached plan of a STATIC edge list, or None.  A list is taken to be static the second time the very same
        tensors (same storage, unchanged in place) are scored: the positive edges of a training loop
        (GripNet-pose.py:137,185), not the negative samples, which are new tensors every epoch.  The cache holds
        the tensors, so their storage cannot be handed to another tensor while an entry is alive.
        (The decision is taken here, in Python, per call: a step captured in a hipGraph replays whatever was decided
        when it was captured.  A buffer that is refilled in place between replays - negative samples - must also be
        refilled between the warm-up calls, or it is taken for static and its plan is replayed on new contents.)"""
        key = (edge_index._version, edge_type._version)
        for k, entry in enumerate(self._seen):
            if entry[0] is edge_index and entry[1] is edge_type and entry[2] == key:
                if entry[3] is None:
                    try:
                        entry[3] = _hip.DistMultPlan(edge_index, edge_type, z.shape[0], self.num_et)
                    except _hip.GripNetHipError:          # too many nodes / relations for the packed encoding
                        entry[3] = False
                self._seen.insert(0, self._seen.pop(k))
                return entry[3] or None
        self._seen.insert(0, [edge_index, edge_type, key, None])
        del self._seen[2:]
        return None

    def forward(self, z, edge_index, edge_type, sigmoid=True):
        _hip.require_gpu(z, edge_index, edge_type, self.weight)
        if z.shape[1] != self.in_dim:
            raise ValueError("expected {} features, got {}".format(self.in_dim, z.shape[1]))
        if recording(z, self.weight):
            plan = self.plan_for(z, edge_index, edge_type)
            if plan is not None and plan.num_nodes != z.shape[0]:
                plan = None
            return DistMultFn.apply(z, self.weight, edge_index, edge_type, sigmoid, plan)
        z = _hip.f32_rows(z)
        out = torch.empty((edge_index.shape[1],), dtype=torch.float32, device=z.device)
        plan = self.plan_for(z, edge_index, edge_type)
        if plan is not None and plan.num_nodes == z.shape[0]:
            try:
                return plan.forward(z, self.weight, sigmoid, out)
            except _hip.GripNetHipError as err:                    # node table too large for the LDS: the general kernels
                if err.status != _hip.GN_ERR_UNSUPPORTED:
                    raise
                self._seen[0][3] = False
        return _hip.distmult(z, edge_index, edge_type, self.weight, sigmoid, out)

    def reset_parameters(self):
        self.weight.data.normal_(std=1 / np.sqrt(self.in_dim))                   # decoder.py:25-26


class multiClassInnerProductDecoder(Module):
    """``softmax(z[node_list] @ W)`` (reference decoder.py:38-45)."""

    def __init__(self, in_dim, num_class):
        super().__init__()
        self.num_class, self.in_dim = num_class, in_dim
        self.weight = Parameter(torch.empty(in_dim, num_class))
        self.reset_parameters()

    def forward(self, z, node_list, softmax=True):
        _hip.require_gpu(z, node_list, self.weight)
        if recording(z, self.weight):
            logits = ClassLogitsFn.apply(z, self.weight, node_list)
            return torch.softmax(logits, dim=1) if softmax else logits
        z = _hip.f32_rows(z)
        nodes = _hip.i64_vec(node_list)
        pred = torch.empty((nodes.shape[0], self.num_class), dtype=torch.float32, device=z.device)
        _hip.gemm(z, self.weight, pred, a_rows=nodes)                            # decoder.py:42
        return _hip.softmax_rows(pred) if softmax else pred

    def reset_parameters(self):
        bound = np.sqrt(6.0 / (self.weight.size(-2) + self.weight.size(-1)))     # decoder.py:47-49
        self.weight.data.uniform_(-bound, bound)
