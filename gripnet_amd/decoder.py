"""Decoders with the reference's names and signatures (gripnet/decoder.py), on the gfx950 kernels.

  multiRelaInnerProductDecoder  decoder.py:10-26   DistMult link scorer
  multiClassInnerProductDecoder decoder.py:29-50   linear + softmax node classifier
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.nn import Module, Parameter

from . import _hip
from .autograd import ClassLogitsFn, DistMultFn, SoftmaxRowsFn, recording


class _EdgeList:
    """One remembered (edge_index, edge_type) pair of multiRelaInnerProductDecoder."""
    __slots__ = ("edge_index", "edge_type", "key", "plan", "copy", "registered")

    def __init__(self, edge_index, edge_type, registered=False):
        self.edge_index, self.edge_type = edge_index, edge_type
        self.key = (edge_index._version, edge_type._version)
        self.plan = None                     # None: not built; False: the packed encoding does not fit; else DistMultPlan
        self.copy = None                     # private copy of the tensors (verify_static)
        self.registered = registered

    def holds(self, edge_index, edge_type):
        return self.edge_index is edge_index and self.edge_type is edge_type


class multiRelaInnerProductDecoder(Module):
    """``sigmoid(sum_k z[u,k] z[v,k] D[r,k])`` (reference decoder.py:19-23).

    Ids outside the tables: the reference's advanced indexing raises IndexError at the call (a device synchronisation per
    call).  Here the kernels write NaN for such an edge and set a per-device flag; the IndexError is raised at the next
    synchronising check: ``gripnet_amd.utils.relation_metrics`` / ``auprc_auroc_ap`` / ``micro_macro`` / ``acc`` (the
    points where the reference's epoch loop reads scores back) or an explicit ``_hip.raise_if_index_errors(device)``.
    Static lists are validated once, when their plan is built, and raise there."""

    # edge lists remembered by plan_for: the reference's epoch loop scores four per epoch (train positives, fresh train
    # negatives, test positives, static test negatives: GripNet-pose.py:137-186)
    CACHE_DEPTH = 6

    def __init__(self, in_dim, num_et):
        super().__init__()
        self.num_et, self.in_dim = num_et, in_dim
        self.weight = Parameter(torch.empty(num_et, in_dim))
        self._seen = []                      # recently scored edge lists (_EdgeList), most recent first
        # auto_static: take a list for static the second time the same, unchanged tensors are scored (see plan_for);
        # False = only lists named with register_static get a plan.
        self.auto_static = True
        # verify_static: before every use of a plan, compare the caller's tensors with a private copy taken when the
        # plan was built and raise if they differ (a full read of the list per call: for callers that refill buffers
        # behind torch's back, e.g. through raw pointers, where `_version` does not move).  GN_VERIFY_STATIC=1 sets it.
        self.verify_static = os.environ.get("GN_VERIFY_STATIC") == "1"
        self.reset_parameters()

    def register_static(self, edge_index, edge_type, num_nodes=None):
        """Promise that this (edge_index, edge_type) pair is a static list (the positive edges of a training loop,
        GripNet-pose.py:137,185): its plan is built on first use - or now, when `num_nodes` is given - whatever
        `auto_static` says, also under stream capture.  The promise holds until forget_static or until the tensors
        are modified through torch (their `_version` moves)."""
        self.__dict__.pop("_memo", None)                     # (recorded shortcuts were decided without the promise)
        entry = self._find(edge_index, edge_type)
        if entry is None:
            entry = _EdgeList(edge_index, edge_type)
            self._insert(entry)
        entry.registered = True
        if entry.key != (edge_index._version, edge_type._version):
            entry.key, entry.plan, entry.copy = (edge_index._version, edge_type._version), None, None
        if num_nodes is not None and entry.plan is None:
            self._build(entry, int(num_nodes))
        return self

    def forget_static(self, edge_index=None, edge_type=None):
        """Drop the plan of one list (or of every list): the next forward scores the raw tensors again."""
        self._seen = [] if edge_index is None else [e for e in self._seen if not e.holds(edge_index, edge_type)]
        self.__dict__.pop("_memo", None)

    def _find(self, edge_index, edge_type):
        for entry in self._seen:
            if entry.holds(edge_index, edge_type):
                return entry
        return None

    def _insert(self, entry):
        self._seen.insert(0, entry)
        while len(self._seen) > self.CACHE_DEPTH:            # one-shot lists (negative samples) go first, built plans last
            victims = [k for k in range(len(self._seen) - 1, 0, -1) if not self._seen[k].plan and not self._seen[k].registered]
            del self._seen[victims[0] if victims else len(self._seen) - 1]

    def _build(self, entry, num_nodes):
        try:
            entry.plan = _hip.DistMultPlan(entry.edge_index, entry.edge_type, num_nodes, self.num_et, self.in_dim)
        except _hip.GripNetHipError:                          # too many nodes / relations for the packed encoding
            entry.plan = False

    def plan_for(self, z, edge_index, edge_type):
        """Cached plan of a STATIC edge list, or None.  A list is static when the caller said so (register_static) or,
        with `auto_static`, the second time the very same tensors (same objects, `_version` unchanged) are scored:
        the positive edges of a training loop (GripNet-pose.py:137,185), not the negative samples, which are new
        tensors every epoch.  The cache holds the tensors, so their storage cannot be handed to another tensor while
        an entry is alive.  Nothing is DECIDED while a stream is being captured: a captured step replays what was
        decided here, so only lists that already have a plan (or were registered) use one inside a capture.
        Writes that bypass torch (raw pointers, another library) do not move `_version`: such callers either switch
        `auto_static` off (and do not register the buffer), or set `verify_static` and get a RuntimeError instead of
        stale scores."""
        if getattr(edge_index, "_gn_volatile", False):        # a buffer the device-side sampler refills in place
            return None
        key = (edge_index._version, edge_type._version)
        entry = self._find(edge_index, edge_type)
        capturing = torch.cuda.is_current_stream_capturing()
        if entry is None:
            if not capturing:
                self._insert(_EdgeList(edge_index, edge_type))
            return None
        if entry.key != key:                                  # modified in place through torch: a new list
            entry.key, entry.plan, entry.copy = key, None, None
            return None
        if entry.plan is None:
            if not entry.registered and (capturing or not self.auto_static):
                return None
            self._build(entry, z.shape[0])
        if entry is not self._seen[0]:
            self._seen.remove(entry)
            self._seen.insert(0, entry)
        if entry.plan and self.verify_static and not capturing:
            if entry.copy is None:
                entry.copy = (edge_index.clone(), edge_type.clone())
            elif not (torch.equal(entry.copy[0], edge_index) and torch.equal(entry.copy[1], edge_type)):
                entry.plan, entry.copy = None, None
                raise RuntimeError("multiRelaInnerProductDecoder: a static edge list changed in place behind its plan "
                                   "(contents differ from the copy taken when the plan was built)")
        return entry.plan or None

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
        if torch.is_grad_enabled() or _hip._recorder is not None or self.verify_static:
            return self._infer(z, edge_index, edge_type, sigmoid, out)
        # steady state (the positive list of an evaluation loop, GripNet-pose.py:185; a buffer the device sampler refills in
        # place): the recorded call (_hip.CallMemo; see layers.homoGraph.forward)
        w = self.weight
        guard = (z.data_ptr(), z.shape[0], z.stride(0), id(edge_index), edge_index._version, id(edge_type), edge_type._version,
                 bool(sigmoid), w.data_ptr(), self.auto_static, _hip.launch_context(z.device), _hip.env_stamp())
        memo = self.__dict__.get("_memo")
        if memo is None:
            memo = self.__dict__["_memo"] = _hip.CallMemo()
        key = guard + (out.data_ptr(),)
        hit = memo.get(key)
        if hit is not None:
            _hip.replay(hit[0])
            return out
        run = lambda: self._infer(z, edge_index, edge_type, sigmoid, out)
        if not memo.second_sighting(guard):
            return run()
        result = memo.record(key, run, drop=(z, out), hold=(edge_index, edge_type))
        entry = self._find(edge_index, edge_type)
        if entry is not None and entry.plan:
            memo.entries[key][1].append(entry.plan)          # (the call names the plan's handle)
        return result

    def _infer(self, z, edge_index, edge_type, sigmoid, out):
        plan = self.plan_for(z, edge_index, edge_type)
        if plan is not None and plan.num_nodes == z.shape[0]:
            try:
                return plan.forward(z, self.weight, sigmoid, out)
            except _hip.GripNetHipError as err:                    # node table too large for the LDS: the general kernels
                if err.status != _hip.GN_ERR_UNSUPPORTED:
                    raise
                self._find(edge_index, edge_type).plan = False
        return _hip.distmult_any(z, edge_index, edge_type, self.weight, sigmoid, out)

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
            return SoftmaxRowsFn.apply(logits) if softmax and logits.shape[0] > 0 else logits
        z = _hip.f32_rows(z)
        nodes = _hip.i64_vec(node_list)
        pred = torch.empty((nodes.shape[0], self.num_class), dtype=torch.float32, device=z.device)
        return _hip.class_scores(z, self.weight, nodes, pred, softmax)           # decoder.py:42-43

    def reset_parameters(self):
        bound = np.sqrt(6.0 / (self.weight.size(-2) + self.weight.size(-1)))     # decoder.py:47-49
        self.weight.data.uniform_(-bound, bound)
