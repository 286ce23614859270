"""Supergraph propagation modules with the reference's names, signatures and state-dict keys,
computed by the gfx950 kernels behind the C ABI (include/gripnet_hip.h).

Mirrors gripnet/layers.py of the reference:
  myGCN      layers.py:15-105   GCN-style conv (internal homogeneous layers, external layer)
  myRGCN     layers.py:108-205  basis-decomposed relational conv with a GLOBAL mean
  homoGraph  layers.py:208-319  internal layer stack of one supervertex
  interGraph layers.py:322-387  external (inter-supervertex) layer

Forward and backward: under ``torch.no_grad()`` (or with nothing that requires grad) a module runs the slot-fused inference
launches, otherwise it routes through the ``torch.autograd.Function``s of autograd.py (same forward kernels, HIP backward).
All tensors must be fp32 / int64 on the GPU (no CPU fallback).  What the reference caches on first use (normalised edge
list) is a *plan* here; the cache protocol (keyed by edge count, RuntimeError on mismatch) is the reference's (layers.py:75-90).
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch.nn import Module, Parameter

from . import _hip
from .autograd import AbsSlotFn, GcnConvFn, HalfSumAbsFn, HalfSumDownFn, RgcnConvFn, Slot, SlotsCatFn, cat_slots, recording


def _unslot(out, side):
    """A frozen layer inside a model that trains (homoGraph / interGraph hand Slots down whenever ANY of their tensors
    requires grad): the inference launches take the Slots' views."""
    if isinstance(out, Slot):
        out = out.view()
    if side is not None and isinstance(side[1], Slot):
        side = (side[0].detach(), side[1].view(), side[2])
    return out, side


def _cat_slots(widths, n_rows, device):
    """Pre-laid-out concat buffer: the layers write their slice in place (layers.py:309,376)."""
    out = torch.empty((n_rows, int(sum(widths))), dtype=torch.float32, device=device)
    views, lo = [], 0
    for w in widths:
        views.append(out[:, lo:lo + w])
        lo += w
    return out, views


class myGCN(Module):
    """``out = A_norm (x W) + b`` (reference layers.py:15-105)."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached = improved, cached
        # "fp32" (the reference's arithmetic) or "bf16": the gathered table x W is rounded to bf16 once and read at
        # half the bytes, everything else stays fp32 (gripnet_amd.utils.set_table_storage; under training since round 6: the
        # forward takes the rounded table, the backward is the fp32 layer's - GcnConvFn)
        self.table_storage = kwargs.get("table_storage", "fp32")
        self.arithmetic = "fp32"          # dense x W: "fp32" (fp32-faithful, default) or "fast" (two-term bf16 splits)
        self.cached_result = None
        self.weight = Parameter(torch.empty(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        bound = math.sqrt(6.0 / (self.weight.size(-2) + self.weight.size(-1)))   # layers.py:43-44
        self.weight.data.uniform_(-bound, bound)
        if self.bias is not None:
            self.bias.data.fill_(0)
        self.cached_result = None
        self.cached_num_edges = None

    @staticmethod
    def norm(edge_index, num_nodes, edge_weight, improved=False, dtype=None):
        """(edge_index', norm) exactly as layers.py:52-69 returns them (built on the GPU)."""
        plan = _hip.GraphPlan.gcn(edge_index, num_nodes, edge_weight, improved)
        ei, nrm = plan.export()
        return ei, (nrm if dtype in (None, torch.float32) else nrm.to(dtype))

    def _plan(self, edge_index, build):
        """The reference's cache protocol (layers.py:75-90): keyed by the edge count only."""
        if self.cached and self.cached_result is not None:
            if edge_index.size(1) != self.cached_num_edges:
                raise RuntimeError("Cached {} number of edges, but found {}".format(
                    self.cached_num_edges, edge_index.size(1)))
        if not self.cached or self.cached_result is None:
            self.cached_num_edges = edge_index.size(1)
            self.cached_result = build()
        return self.cached_result

    def _run(self, plan, x, n_out, out, relu, side, planes=None, passthrough=False):
        self._planes_written = False                 # did this call's launch leave the split planes it was offered?
        if recording(x, self.weight, self.bias):                                 # training: autograd path (out / side: Slots)
            y = GcnConvFn.apply(x, self.weight, self.bias, plan, n_out, relu, out, side, planes, passthrough, self.table_storage)
            self._planes_written = planes is not None
            return y
        if passthrough:                                                          # (a frozen layer: x goes to the concat as it is)
            return self._run(plan, x, n_out, out, relu, side, planes), x
        out, side = _unslot(out, side)
        x = _hip.f32_rows(x.detach())
        if out is None:
            out = torch.empty((n_out, self.out_channels), dtype=torch.float32, device=x.device)
        if self.table_storage == "bf16" and self.out_channels % 8 == 0 and _hip.ld(out) % 4 == 0 and out.data_ptr() % 16 == 0:
            # layers.py:73 in fp32 arithmetic; the product's own store rounds the table to bf16 (round 6: no gn_cast_bf16 pass),
            # shapes outside the tall-skinny kernel leave a fp32 product that aggregate_bf16 rounds
            xw = torch.empty((x.shape[0], self.out_channels), dtype=torch.bfloat16, device=x.device)
            try:
                _hip.gemm(x, self.weight, xw, out_bf16=True, fast=self.arithmetic == "fast")
            except _hip.GripNetHipError as err:
                if err.status != _hip.GN_ERR_UNSUPPORTED:
                    raise
                xw = torch.empty((x.shape[0], self.out_channels), dtype=torch.float32, device=x.device)
                _hip.gemm(x, self.weight, xw, fast=self.arithmetic == "fast")
            return plan.aggregate_bf16(xw, self.bias, relu, out, side)
        if self.weight.is_contiguous() and (plan.blocked_ok(self.in_channels, self.out_channels, x) or
                                            plan.transform_ok(self.in_channels, self.out_channels, x)):
            # A_norm (x W) = (A_norm x) W: the contraction of layers.py:73 runs on the aggregated row
            y = plan.aggregate(x, self.bias, relu, out, side, weight=self.weight, planes=planes)
            self._planes_written = planes is not None
            return y
        xw = torch.empty((x.shape[0], self.out_channels), dtype=torch.float32, device=x.device)
        _hip.gemm(x, self.weight, xw, fast=self.arithmetic == "fast")               # layers.py:73
        self._planes_written = planes is not None
        return plan.aggregate(xw, self.bias, relu, out, side, planes=planes)     # layers.py:92-100

    def forward(self, x, edge_index, edge_weight=None, *, _out=None, _relu=False, _side=None, _pass=False):
        # (_pass: also return x, for the concat that holds it - homoGraph's training path; GcnConvFn)
        _hip.require_gpu(x, edge_index, edge_weight, self.weight)
        n = x.size(0)
        def build():
            plan = _hip.GraphPlan.gcn(edge_index, n, edge_weight, self.improved)
            if self.cached:                              # LDS-staged gathers where the graph qualifies: a host-side
                plan.build_blocked(self.out_channels)    # schedule of the CSR, worth it only for a graph that is kept
            return plan
        plan = self._plan(edge_index, build)
        return self._run(plan, x, n, _out, _relu, _side, passthrough=_pass)

    def forward_bipartite(self, x, inter_edge_index, n_target, edge_weight=None, *, _out=None, _relu=False, _side=None,
                          _planes=None):
        """The conv as interGraph uses it (layers.py:363-368), in closed form: rows are targets."""
        _hip.require_gpu(x, inter_edge_index, edge_weight, self.weight)
        n_src = x.size(0)
        plan = self._plan(inter_edge_index,
                          lambda: _hip.GraphPlan.bipartite(inter_edge_index, n_src, n_target, edge_weight))
        return self._run(plan, x, n_target, _out, _relu, _side, _planes)

    def __repr__(self):
        return "{}({}, {})".format(self.__class__.__name__, self.in_channels, self.out_channels)


class myRGCN(Module):
    """``out[i] = mean_{e: dst=i} x[src_e] W_{r(e)} + x[i] root (+ b)``, ``W_r = sum_b att[r,b] basis[b]``
    with the mean over ALL incoming edges of all relations (reference layers.py:108-205)."""

    def __init__(self, in_channels, out_channels, num_relations, num_bases, after_relu, bias=False, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations, self.num_bases, self.after_relu = num_relations, num_bases, after_relu
        self.basis = Parameter(torch.empty(num_bases, in_channels, out_channels))
        self.att = Parameter(torch.empty(num_relations, num_bases))
        self.root = Parameter(torch.empty(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._plan = None
        self._plan_key = None
        # Arithmetic of the dense products (an argument of the C ABI, never the environment): "fp32" (default) is
        # fp32-faithful - three-term bf16 splits or the fp32 matrix instruction, the reference's precision
        # (layers.py:172-186 are fp32 matmuls); "fast" keeps two bf16 terms (<= 2^-16 per product).  `kernel` picks a
        # relational kernel for tests and measurements ("auto": the library decides).
        self.arithmetic = "fp32"
        self.kernel = "auto"
        # static_graph = True (default): the layer's edge list is the same tensor call after call (a training loop's
        # train_idx, GripNet-pose.py:121-127) and gets the full plan - every kernel's schedule, ~15 ms once on PoSE-0.
        # False: edge lists change from call to call; each gets a LIGHT plan (the device-sorted key list only, a fraction of
        # the build) and runs on the general O(E) kernel - closer to the reference's myRGCN, which has no set-up cost at all
        # (layers.py:165-169).
        self.static_graph = True
        self.reset_parameters()

    def _fast(self):
        if self.arithmetic not in ("fp32", "fast"):
            raise ValueError("arithmetic must be 'fp32' or 'fast', got {!r}".format(self.arithmetic))
        return self.arithmetic == "fast"

    def reset_parameters(self):
        self.att.data.normal_(std=1 / np.sqrt(self.num_bases))                   # layers.py:152
        std = 2 / self.in_channels if self.after_relu else 1 / np.sqrt(self.in_channels)
        self.root.data.normal_(std=std)
        self.basis.data.normal_(std=std)
        if self.bias is not None:
            self.bias.data.zero_()

    def plan_for(self, edge_index, range_list, num_nodes, edge_lo=None, edge_hi=None):
        """Plan of the static relational graph, rebuilt when another edge tensor or range list arrives or one of them
        is modified in place.  The cache entry holds the caller's objects (compared with `is`), so an address or id
        reused by a later tensor cannot pass for the old one."""
        key = (edge_index._version, getattr(range_list, "_version", 0), num_nodes, edge_lo, edge_hi, bool(self.static_graph))
        held = self._plan_key
        if (self._plan is None or held is None or held[0] is not edge_index or held[1] is not range_list or held[2] != key):
            self._plan = _hip.RgcnPlan(edge_index, range_list, num_nodes, edge_lo, edge_hi, light=not self.static_graph)
            self._plan_key = (edge_index, range_list, key)
        return self._plan

    def start_pair_sums(self, edge_index, range_list, num_nodes):
        """OPTIONAL, once per step and before the layers that produce this layer's input: starts the x-INDEPENDENT half of
        the layer - the att rows of the edges of every (destination, source) pair, summed (rgcn_pair.hip; W_r is never
        formed, layers.py:172-173 becomes these sums) - on the device's side stream, so that it runs beside the gene and
        external layers instead of in front of this layer's contraction.  The next inference `forward` on the same graph
        then only contracts the sums with x (same bits as the one-launch form).  The sums are recomputed by every call -
        nothing is kept across steps.  Returns False (and does nothing) where the split does not apply: another kernel
        than the destination-major one, widths whose x does not travel as split planes, arithmetic "fast", training."""
        _hip.require_gpu(edge_index, self.att)
        self._sums_pending = None
        if torch.is_grad_enabled() and recording(self.basis, self.att, self.root, self.bias):
            return False
        if self.in_channels % 16 != 0 or self._fast() or self.kernel not in ("auto", "pair"):
            return False
        plan = self.plan_for(edge_index, range_list, num_nodes)
        if plan.path(self.in_channels, self.out_channels, self.num_bases, False, self.kernel) != "pair":
            return False
        plan.start_pair_sums(self.att.detach(), self.in_channels, self.out_channels, self.num_bases, _hip.side_stream(edge_index.device))
        self._sums_pending = plan
        return True

    def _take_pair_sums(self, plan, planes):
        """Are this step's pair sums on their way for `plan`?  (Consumes the note start_pair_sums left.)"""
        pend, self._sums_pending = self.__dict__.get("_sums_pending"), None
        return pend is not None and pend is plan and planes is not None and not self._fast()

    def forward(self, x, edge_index, edge_type, range_list, *, _out=None, _relu=False, _side=None, _pass=False):
        # edge_type is accepted and unused, as in the reference (the relation of an edge is the
        # range_list row that contains it, layers.py:171-186)
        _hip.require_gpu(x, edge_index, self.basis)
        x = _hip.f32_rows(x)
        if x.shape[1] != self.in_channels:
            raise ValueError("expected {} input features, got {}".format(self.in_channels, x.shape[1]))
        if range_list.shape[0] != self.num_relations:
            raise ValueError("range_list has {} rows for {} relations".format(range_list.shape[0], self.num_relations))
        plan = self.plan_for(edge_index, range_list, x.shape[0])
        if recording(x, self.basis, self.att, self.root, self.bias):             # training: autograd path
            planes = _hip.SplitPlanes.of(x, self.in_channels // 16) if self.in_channels % 16 == 0 else None
            return RgcnConvFn.apply(x, self.basis, self.att, self.root, self.bias, plan, _relu, _out, _side, planes, _pass)   # (Slots)
        if _pass:                                                                # (a frozen layer: x goes to the concat as it is)
            return self.forward(x, edge_index, edge_type, range_list, _out=_out, _relu=_relu, _side=_side), x
        _out, _side = _unslot(_out, _side)
        out = _out if _out is not None else torch.empty((x.shape[0], self.out_channels), dtype=torch.float32,
                                                        device=x.device)
        planes = _hip.SplitPlanes.of(x, self.in_channels // 16) if self.in_channels % 16 == 0 else None
        return plan.forward(x, self.basis, self.att, self.root, self.bias, _relu, out, side=_side, fast=self._fast(),
                            path=self.kernel, x_planes=planes, pair_sums=self._take_pair_sums(plan, planes))

    def __repr__(self):
        return "{}({}, {}, num_relations={})".format(self.__class__.__name__, self.in_channels,
                                                     self.out_channels, self.num_relations)


class homoGraph(Module):
    """Internal layer stack of one supervertex (reference layers.py:208-319):
    ``h_l = relu(conv_l(h_{l-1}))`` for every layer, optional concat of the input and all layers."""

    def __init__(self, nhid_list, requires_grad=True, start_graph=False, in_dim=None, multi_relational=False,
                 n_rela=None, n_base=32):
        super().__init__()
        self.multi_relational = multi_relational
        self.start_graph = start_graph
        self.out_dim = nhid_list[-1]
        self.n_cov = len(nhid_list) - 1
        self.nhid_list = list(nhid_list)
        if start_graph:
            self.embedding = Parameter(torch.empty(in_dim, nhid_list[0]))
            self.embedding.requires_grad = requires_grad
            self.reset_parameters()
        if multi_relational:
            assert n_rela is not None
            self.conv_list = torch.nn.ModuleList([
                myRGCN(nhid_list[i], nhid_list[i + 1], n_rela, n_base, after_relu=i > 0)
                for i in range(len(nhid_list) - 1)])
        else:
            self.conv_list = torch.nn.ModuleList([
                myGCN(nhid_list[i], nhid_list[i + 1], cached=True) for i in range(len(nhid_list) - 1)])

    def reset_parameters(self):
        self.embedding.data.normal_()

    def forward(self, x, homo_edge_index, edge_weight=None, edge_type=None, range_list=None, if_catout=False):
        if self.start_graph:
            x = self.embedding                                                  # layers.py:261-262
        if self.multi_relational:
            assert edge_type is not None
            assert range_list is not None
        _hip.require_gpu(x, homo_edge_index)
        if torch.is_grad_enabled() and recording(x, *self.parameters()):         # training: autograd-tracked
            # the concat of layers.py:309 without a copy: every layer writes its columns of one buffer, the first layer's launch
            # copies the input into its columns (as on the inference path); SlotsCatFn hands the gradient's columns back
            slots = [None] * (len(self.conv_list) + 1)
            if if_catout:
                _, slots = cat_slots([x.shape[1]] + [c.out_channels for c in self.conv_list], x.shape[0], x.device)
            # (with the concat a layer hands its input on to it - `_pass` - so that the input's gradient is ONE sum, made where
            # the layer's backward stores dx, not a launch of the autograd engine's)
            outs, h = [], x
            for i, net in enumerate(self.conv_list):
                side = (x, slots[0], 0) if if_catout and i == 0 else None
                if self.multi_relational:
                    h = net(h, homo_edge_index, edge_type, range_list, _relu=True, _out=slots[i + 1], _side=side, _pass=if_catout)
                else:
                    h = net(h, homo_edge_index, edge_weight, _relu=True, _out=slots[i + 1], _side=side, _pass=if_catout)
                if if_catout:
                    h, seen = h
                    outs.append(seen)
            return SlotsCatFn.apply(slots, *outs, h) if if_catout else h
        x = _hip.f32_rows(x)
        n = x.shape[0]
        widths = [x.shape[1]] + [c.out_channels for c in self.conv_list]
        # the output is a fresh tensor on every call (as the reference's); with if_catout the layers write their columns of it
        out = torch.empty((n, sum(widths) if if_catout else widths[-1]), dtype=torch.float32, device=x.device)
        if torch.is_grad_enabled() or _hip._recorder is not None:
            return self._infer(x, homo_edge_index, edge_weight, edge_type, range_list, if_catout, widths, out)
        # steady state: the same graph tensors, the same input and output addresses -> the recorded calls (_hip.CallMemo)
        convs = self.conv_list
        planes = _hip.SplitPlanes.of(x, widths[0] // 16) if self.multi_relational and widths[0] % 16 == 0 else None
        guard = (x.data_ptr(), n, widths[0], x.stride(0), id(homo_edge_index), homo_edge_index._version,
                 id(edge_weight), 0 if edge_weight is None else edge_weight._version,
                 id(range_list), getattr(range_list, "_version", 0), if_catout, id(planes),
                 tuple([(id(c._plan) if self.multi_relational else id(c.cached_result), c.arithmetic,
                         c.kernel if self.multi_relational else c.table_storage, c.__dict__.get("_sums_pending") is not None) +
                        tuple([0 if p is None else p.data_ptr() for p in c._parameters.values()]) for c in convs]),
                 _hip.launch_context(x.device), _hip.env_stamp())
        memo = self.__dict__.get("_memo")
        if memo is None:
            memo = self.__dict__["_memo"] = _hip.CallMemo()
        key = guard + (out.data_ptr(),)
        hit = memo.get(key)
        if hit is not None:
            _hip.replay(hit[0])
            if self.multi_relational:
                for c in convs:                                  # (the replayed calls consumed this step's pair sums)
                    c._sums_pending = None
            return out
        run = lambda: self._infer(x, homo_edge_index, edge_weight, edge_type, range_list, if_catout, widths, out)
        if not memo.second_sighting(guard):
            return run()
        result = memo.record(key, run, drop=(x, out), hold=(homo_edge_index, edge_weight, range_list, planes))
        # (the key names the plans the recording ran on by id: the entry holds them, so that an id cannot be reused)
        memo.entries[key][1].extend([c._plan if self.multi_relational else c.cached_result for c in convs])
        return result

    def _infer(self, x, homo_edge_index, edge_weight, edge_type, range_list, if_catout, widths, out):
        """The inference launches: conv + ReLU fused, every layer; the last layer (every layer and the input, with
        if_catout) writes into `out`."""
        side = None
        if if_catout:
            slots, lo = [], 0
            for w in widths:
                slots.append(out[:, lo:lo + w])
                lo += w
            side = (x, slots[0], 0)                          # slot 0 <- input, copied by the first layer's launch
        else:
            slots = [None] * len(self.conv_list) + [out]
        h = x
        for i, net in enumerate(self.conv_list):                                 # conv + ReLU fused, every layer
            if self.multi_relational:
                h = net(h, homo_edge_index, edge_type, range_list, _out=slots[i + 1], _relu=True, _side=side)
            else:
                h = net(h, homo_edge_index, edge_weight, _out=slots[i + 1], _relu=True, _side=side)
            side = None
        return out


class interGraph(Module):
    """External layer source supervertex -> target supervertex (reference layers.py:322-387)."""

    def __init__(self, source_dim, target_dim, n_target, target_feat_dim=32, requires_grad=True,
                 if_one_external=True):
        super().__init__()
        self.source_dim, self.target_dim = source_dim, target_dim
        self.target_feat_dim, self.n_target = target_feat_dim, n_target
        self.if_one_external = if_one_external
        if not self.if_one_external:
            self.conv = myGCN(source_dim, target_dim, cached=True)
            return
        self.target_feat = Parameter(torch.empty(n_target, target_feat_dim))
        self.target_feat.requires_grad = requires_grad
        if target_dim != target_feat_dim:                                        # exists even if unused
            self.target_feat_down = Parameter(torch.empty(target_feat_dim, target_dim))
            self.target_feat_down.requires_grad = requires_grad
            self.target_feat_down.data.normal_()
        self.conv = myGCN(source_dim, target_dim, cached=True)
        self.reset_parameters()

    def reset_parameters(self):
        self.target_feat.data.normal_()

    def forward(self, x, inter_edge_index, edge_weight=None, if_relu=True, mod="cat"):
        _hip.require_gpu(x, inter_edge_index)
        dev = x.device
        if torch.is_grad_enabled() and recording(x, *self.parameters()):                                     # training: autograd-tracked glue
            if self.if_one_external and mod == "cat":
                # [y | |target_feat|] (layers.py:376) in one launch: both written into their columns, no concat, no abs
                _, (ys, ts) = cat_slots([self.target_dim, self.target_feat_dim], self.n_target, dev)
                # the same launch leaves the row as bf16 split planes (as on the inference path): a relational layer that takes
                # this output contracts with them instead of splitting x in every unit
                width, planes = self.target_dim + self.target_feat_dim, None
                if width % 16 == 0 and width <= 64 and self.conv.table_storage == "fp32":
                    planes = getattr(self, "_planes", None)
                    if planes is None or planes.device != dev:
                        planes = self._planes = _hip.SplitPlanes(self.n_target, width // 16, dev)
                y = self.conv.forward_bipartite(x, inter_edge_index, self.n_target, edge_weight, _relu=if_relu, _out=ys,
                                                _side=(self.target_feat, ts, 1),
                                                _planes=None if planes is None else (planes, 0, self.target_dim))
                out = SlotsCatFn.apply([ys, ts], y, AbsSlotFn.apply(self.target_feat, ts))
                if planes is not None and getattr(self.conv, "_planes_written", False):
                    planes.tag(out)
                return out
            y = self.conv.forward_bipartite(x, inter_edge_index, self.n_target, edge_weight, _relu=if_relu)
            if not self.if_one_external:
                return y
            if y.shape[1] == self.target_feat.shape[1]:                          # layers.py:378-379
                return HalfSumAbsFn.apply(y, self.target_feat)
            return HalfSumDownFn.apply(y, self.target_feat, self.target_feat_down)   # layers.py:381-384
        if not self.if_one_external:                                             # layers.py:372-373
            return self.conv.forward_bipartite(x, inter_edge_index, self.n_target, edge_weight, _relu=if_relu)
        if mod == "cat":                                                         # layers.py:375-376
            x = _hip.f32_rows(x)
            out = torch.empty((self.n_target, self.target_dim + self.target_feat_dim), dtype=torch.float32, device=dev)
            conv, tf = self.conv, self.target_feat
            if torch.is_grad_enabled() or _hip._recorder is not None or not tf.is_contiguous():
                return self._infer_cat(x, inter_edge_index, edge_weight, if_relu, out)
            # steady state: the recorded calls (_hip.CallMemo; see homoGraph.forward)
            guard = (x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), id(inter_edge_index), inter_edge_index._version,
                     id(edge_weight), 0 if edge_weight is None else edge_weight._version, bool(if_relu),
                     id(conv.cached_result), conv.table_storage, conv.arithmetic, tf.data_ptr(),
                     tuple([0 if p is None else p.data_ptr() for p in conv._parameters.values()]),
                     _hip.launch_context(dev), _hip.env_stamp())
            memo = self.__dict__.get("_memo")
            if memo is None:
                memo = self.__dict__["_memo"] = _hip.CallMemo()
            key = guard + (out.data_ptr(),)
            hit = memo.get(key)
            if hit is not None:
                _hip.replay(hit[0])
                planes = hit[2]
                if planes is not None:                       # the launch rewrote the split planes: they describe THIS output now
                    planes.generation += 1
                    planes.tag(out)
                return out
            run = lambda: self._infer_cat(x, inter_edge_index, edge_weight, if_relu, out)
            if not memo.second_sighting(guard):
                return run()
            result = memo.record(key, run, drop=(x, out), hold=(inter_edge_index, edge_weight))
            entry = memo.entries[key]
            entry[1].append(conv.cached_result)
            memo.entries[key] = (entry[0], entry[1], _hip.SplitPlanes.of(out, (self.target_dim + self.target_feat_dim) // 16)
                                 if (self.target_dim + self.target_feat_dim) % 16 == 0 else None)
            return result
        y = self.conv.forward_bipartite(x, inter_edge_index, self.n_target, edge_weight, _relu=if_relu)
        if y.shape[1] == self.target_feat.shape[1]:                              # layers.py:378-379
            return _hip.merge(y, self.target_feat, 2)
        down = torch.empty_like(y)                                               # layers.py:381-384
        _hip.gemm(self.target_feat, self.target_feat_down, down)
        return _hip.merge(y, down, 3)

    def _infer_cat(self, x, inter_edge_index, edge_weight, if_relu, out):
        """[y | |target_feat|] (layers.py:375-376) written into the columns of `out` by ONE launch, which also leaves the
        row as bf16 split planes: a relational layer that takes this output as its input (GripNet-pose.py:120-127)
        contracts with them instead of splitting x in every unit."""
        dev = x.device
        y, tf = out[:, :self.target_dim], out[:, self.target_dim:]
        width, planes = self.target_dim + self.target_feat_dim, None
        if width % 16 == 0 and width <= 64 and self.conv.table_storage == "fp32":
            planes = getattr(self, "_planes", None)
            if planes is None or planes.device != dev:
                planes = self._planes = _hip.SplitPlanes(self.n_target, width // 16, dev)
        self.conv.forward_bipartite(x, inter_edge_index, self.n_target, edge_weight, _out=y, _relu=if_relu,
                                    _side=(self.target_feat, tf, 1),     # |target_feat| slot, same launch
                                    _planes=None if planes is None else (planes, 0, self.target_dim))
        if planes is not None and getattr(self.conv, "_planes_written", False):
            planes.tag(out)
        return out
