"""Counterparts of the reference drivers' model assembly and forward call sequence.

The reference builds its encoder inline in every script (``Model(gg, gd, dd, dmt)`` whose own
``forward`` is ``pass``, GripNet-pose.py:73-99) and spells the call order out in ``train()``
(GripNet-pose.py:117-138).  This module is that harness for the build's own benchmarks and
tests: same hyper-parameters, same call order and keyword arguments, same state-dict keys.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch.nn import Module

from . import _hip
from .decoder import multiClassInnerProductDecoder, multiRelaInnerProductDecoder
from .layers import homoGraph, interGraph

POSE_HPARAMS = dict(gg_nhids=[32, 16, 16], gd_out=[16, 32])     # GripNet-pose.py:86-89


class PoseModel(Module):
    """gg (gene internal) -> gd (external) -> dd (drug internal, relational) -> dmt (DistMult)."""

    def __init__(self, n_g_node, n_d_node, n_dd_edge_type, gg_nhids=None, gd_out=None, n_base=32):
        super().__init__()
        gg_nhids = list(gg_nhids or POSE_HPARAMS["gg_nhids"])
        gd_out = list(gd_out or POSE_HPARAMS["gd_out"])
        dd_nhids = [sum(gd_out), 32]
        self.gg = homoGraph(gg_nhids, start_graph=True, in_dim=n_g_node)                       # pose.py:95
        self.gd = interGraph(sum(gg_nhids), gd_out[0], n_d_node, target_feat_dim=gd_out[-1])   # pose.py:96
        self.dd = homoGraph(dd_nhids, multi_relational=True, n_rela=n_dd_edge_type, n_base=n_base)  # pose.py:97
        self.dmt = multiRelaInnerProductDecoder(sum(dd_nhids), n_dd_edge_type)                 # pose.py:98

    # True: every inference step starts the relational layer's x-independent half (the pair sums of its att rows) on the side
    # stream before the gene layers (myRGCN.start_pair_sums) and the layer itself only contracts them.  Measured in round 6
    # (profiles/r06_pair_sums.md); the default is what measured faster.
    split_relational = False

    def encode(self, data):
        if self.split_relational and not torch.is_grad_enabled():
            self.dd.conv_list[0].start_pair_sums(data.train_idx, data.train_range, int(data.n_d_node))
        z = self.gg(None, data.gg_edge_index, edge_weight=data.edge_weight, if_catout=True)    # pose.py:117-119
        z = self.gd(z, data.gd_edge_index, mod="cat", if_relu=True)                            # pose.py:120
        return self.dd(z, data.train_idx, edge_type=data.train_et, range_list=data.train_range,
                       if_catout=True)                                                         # pose.py:121-127

    def forward(self, data, sigmoid=True):
        z = self.encode(data)
        score = self.dmt(z, data.train_idx, data.train_et, sigmoid=sigmoid)                    # pose.py:137
        return z, score


class AminerModel(Module):
    """pp -> pa -> aa -> mcip (GripNet-aminer.py:96-108,124-130; freebase-b is the same shape)."""

    def __init__(self, n_p_node, n_a_node, n_class, pp_nhids=(128, 64, 64), pa_out=(64, 64), aa_hidden=(128, 32)):
        super().__init__()
        pp_nhids, pa_out = list(pp_nhids), list(pa_out)
        aa_nhids = [sum(pa_out)] + list(aa_hidden)
        self.pp = homoGraph(pp_nhids, start_graph=True, in_dim=n_p_node)
        self.pa = interGraph(sum(pp_nhids), pa_out[0], n_a_node, target_feat_dim=pa_out[-1])
        self.aa = homoGraph(aa_nhids)
        self.mcip = multiClassInnerProductDecoder(sum(aa_nhids), n_class)

    def forward(self, data, node_list, softmax=True):
        z = self.pp(None, data.pp_edge_idx, edge_weight=data.pp_edge_weight, if_catout=True)
        z = self.pa(z, data.pa_edge_idx, if_relu=True, mod="cat")
        z = self.aa(z, data.aa_edge_idx, edge_weight=data.aa_edge_weight, if_catout=True)
        return z, self.mcip(z, node_list, softmax=softmax)


class FreebaseAModel(Module):
    """pp -> mcip on ONE supervertex, no concat (GripNet-freebase-a.py:94,101-104,120-122)."""

    def __init__(self, n_a_node, n_class, pp_nhids=(256, 128, 128)):
        super().__init__()
        pp_nhids = list(pp_nhids)
        self.pp = homoGraph(pp_nhids, start_graph=True, in_dim=n_a_node)
        self.mcip = multiClassInnerProductDecoder(pp_nhids[-1], n_class)

    def forward(self, data, node_list, softmax=True):
        z = self.pp(None, data.aa_edge_idx, edge_weight=data.aa_edge_weight)       # freebase-a.py:120 (if_catout off)
        return z, self.mcip(z, node_list, softmax=softmax)


class FreebaseBModel(AminerModel):
    """The aminer call sequence with freebase-b's widths (GripNet-freebase-b.py:96-98: pa_out = [128, 128])."""

    def __init__(self, n_p_node, n_a_node, n_class, pp_nhids=(128, 64, 64), pa_out=(128, 128), aa_hidden=(128, 32)):
        super().__init__(n_p_node, n_a_node, n_class, pp_nhids=pp_nhids, pa_out=pa_out, aa_hidden=aa_hidden)


class FreebaseCModel(Module):
    """pp -> pa, qq -> qa, (z + z1 + aa_embeddings) / 3 -> aa -> mcip
    (GripNet-freebase-c.py:102-136,150-165; freebase-d is the same shape)."""

    def __init__(self, n_p_node, n_q_node, n_a_node, n_class, pp_nhids=(256, 128, 128), qq_nhids=(256, 128, 128),
                 pa_out=(128, 128), aa_hidden=(32,)):
        super().__init__()
        pp_nhids, qq_nhids, pa_out = list(pp_nhids), list(qq_nhids), list(pa_out)
        aa_nhids = [pa_out[-1]] + list(aa_hidden)
        self.pp = homoGraph(pp_nhids, start_graph=True, in_dim=n_p_node)
        self.pa = interGraph(sum(pp_nhids), pa_out[0], n_a_node, target_feat_dim=pa_out[-1], if_one_external=False)
        self.qq = homoGraph(qq_nhids, start_graph=True, in_dim=n_q_node)
        self.qa = interGraph(sum(qq_nhids), pa_out[0], n_a_node, target_feat_dim=pa_out[-1], if_one_external=False)
        self.aa_embeddings = torch.nn.Parameter(torch.empty(n_a_node, aa_nhids[0]).normal_())
        self.aa = homoGraph(aa_nhids)
        self.mcip = multiClassInnerProductDecoder(aa_nhids[-1], n_class)

    def forward(self, data, node_list, softmax=True):
        z = self.pp(None, data.pp_edge_idx, edge_weight=data.pp_edge_weight, if_catout=True)
        z = self.pa(z, data.pa_edge_idx, mod="add", if_relu=True)
        z1 = self.qq(None, data.qq_edge_idx, edge_weight=data.qq_edge_weight, if_catout=True)
        z1 = self.qa(z1, data.qa_edge_idx, mod="add", if_relu=True)
        if torch.is_grad_enabled() and (z.requires_grad or self.aa_embeddings.requires_grad):
            from .autograd import MergeMeanFn
            merged = MergeMeanFn.apply(z, z1, self.aa_embeddings)                # training: autograd-tracked, the library's launches
        else:
            merged = _hip.merge(z, z1, 4, src2=self.aa_embeddings)               # (z + z1 + aae) / 3
        z = self.aa(merged, data.aa_edge_idx, edge_weight=data.aa_edge_weight)
        return z, self.mcip(z, node_list, softmax=softmax)


class RgcnPoseModel(Module):
    """embedding -> myRGCN -> myRGCN -> DistMult over ALL nodes of the homogenised graph: the reference's improved-RGCN
    baseline (baselines/LP_baselines/rgcn_pose.py:53-77 model, :91-106 call order; `sparse_id(n) @ embedding` is the
    embedding itself).  The relational layers run far beyond the LDS-resident kernels (N ~ 2 x 10^4, R ~ 10^3)."""

    def __init__(self, n_node, n_edge_type, dims=(64, 32, 32), n_bases=16):
        super().__init__()
        from .layers import myRGCN
        self.embedding = torch.nn.Parameter(torch.empty(n_node, dims[0]).normal_())            # rgcn_pose.py:64-65
        self.rgcn1 = myRGCN(dims[0], dims[1], n_edge_type, n_bases, after_relu=False)           # rgcn_pose.py:73
        self.rgcn2 = myRGCN(dims[1], dims[2], n_edge_type, n_bases, after_relu=True)            # rgcn_pose.py:74
        self.dmt = multiRelaInnerProductDecoder(dims[2], n_edge_type)                           # rgcn_pose.py:75

    def encode(self, data):
        z = self.rgcn1(self.embedding, data.train_idx, data.train_et, data.train_range)        # rgcn_pose.py:95-97
        return self.rgcn2(z, data.train_idx, data.train_et, data.train_range)

    def forward(self, data, sigmoid=True):
        z = self.encode(data)
        return z, self.dmt(z, data.train_idx, data.train_et, sigmoid=sigmoid)                   # rgcn_pose.py:110


def load_reference_state(model: Module, state: Dict[str, torch.Tensor], strict: bool = True):
    """Load a state dict saved by the reference (GripNet-pose.py:236; keys of SURVEY App. B.2)."""
    return model.load_state_dict(state, strict=strict)


class Graphed:
    """A launch-bound stage captured once into a hipGraph and replayed (torch.cuda.CUDAGraph on
    ROCm).  ``fn`` must be free of host synchronisation and read only static tensors; the library's
    entry points launch on torch's current stream, so they are captured like any torch op."""

    def __init__(self, fn):
        self.fn, self.graph, self.out = fn, None, None

    def capture(self):
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # plans / workspaces are created here, not in the capture
            for _ in range(2):
                self.fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self.fn()
        return self

    def __call__(self):
        self.graph.replay()
        return self.out


class Recorded:
    """A launch-bound stage recorded once as the list of C-ABI calls it makes (gripnet_amd._hip.Recorder) and replayed by
    making those calls again from one loop: ordinary launches on the stream they were recorded on, on the buffers they
    were recorded with.  Unlike a hipGraph replay, which on this stack starts ~9 us after the stream's previous work has
    finished, they queue up behind it - and unlike the eager path they do not pay the Python layers above every entry
    point.  `fn` must be free of host synchronisation, torch kernels and data-dependent Python decisions (the library's
    entry points only), and must be replayed with the stream current that was current when it was recorded."""

    def __init__(self, fn):
        self.fn, self.calls, self.out, self._rec = fn, None, None, None

    def capture(self):
        for _ in range(2):                            # plans exist, the decoder has seen its list twice
            self.fn()
        torch.cuda.synchronize()
        with _hip.Recorder() as rec:
            self.out = self.fn()
        self._rec, self.calls = rec, rec.calls        # (the recorder holds the operands of every call)
        return self

    def __call__(self):
        _hip.replay(self.calls)
        return self.out


class PoseStages:
    """The steady-state forward of PoseModel cut into three stages with static buffers, so that the
    launch-bound ones can be replayed as hipGraphs and the relational layer can be bracketed by HIP
    events on its own (bench.py's roofline):

        genes()   gg -> gd                         -> x      [n_d, 48]
        drugs()   dd: concat slot 0 + myRGCN       -> z      [n_d, 80]
        decode()  DistMult on the positive edges   -> score  [E]

    Same calls and keyword arguments as PoseModel.forward / GripNet-pose.py:117-138.
    """

    def __init__(self, model: PoseModel, data, graphs: bool = True, edge_index=None, edge_type=None,
                 timed_entry: Optional[str] = None, recorded: bool = False):
        """`timed_entry` ("gn_rgcn_forward_f32" or "gn_distmult[_plan]_forward_f32"): with graphs, the stage that holds
        that entry point is NOT captured (nor, behind a timed relational layer, the one-kernel decoder): its launches are made from Python, so that an active
        _hip.KernelTimer brackets the entry point itself with HIP events (events around a graph replay would add the
        graph's launch latency, ~7 us, to the measured duration); every other stage is a hipGraph."""
        self.model, self.data = model, data
        self.conv = model.dd.conv_list[0]
        dev = data.train_idx.device
        n_d = int(data.n_d_node)
        self.idx = data.train_idx if edge_index is None else edge_index      # decoder's edge list (a shard's slice)
        self.et = data.train_et if edge_type is None else edge_type
        self.z = None                                # [n_d, 80]: the drug stack's output of the last step
        self._genes, self._drugs, self._decode = self._genes_eager, self._drugs_eager, self._decode_eager
        self._encode = None                          # graphs, decoder timed or nothing timed: genes and drugs as ONE graph
        self.timed_entry, self.graphs = timed_entry, graphs
        self.x = None
        self._whole = None                           # recorded: the whole step as one list of entry-point calls
        if recorded:
            with torch.no_grad():
                self._whole = Recorded(self._step_eager).capture()
        elif graphs:
            with torch.no_grad():
                if timed_entry == "gn_rgcn_forward_f32":
                    self._genes = Graphed(self._genes_eager).capture()
                    self.x = self._genes()
                    self._drugs_eager()                      # builds the plan; stays eager
                    for _ in range(2):                       # the decoder is one launch: as a graph of its own it
                        self._decode_eager()                 # would cost a graph launch (~7 us) instead of a kernel launch
                else:
                    self._encode = Graphed(self._encode_eager).capture()
                    if timed_entry not in ("gn_distmult_forward_f32", "gn_distmult_plan_forward_f32"):
                        self._decode = Graphed(self._decode_eager).capture()
                    else:
                        for _ in range(2):                   # second sighting of the edge list: the decoder's plan
                            self._decode_eager()

    def _encode_eager(self):
        self.x = self._genes_eager()
        return self._drugs_eager()

    def _genes_eager(self):
        d = self.data
        z = self.model.gg(None, d.gg_edge_index, edge_weight=d.edge_weight, if_catout=True)
        return self.model.gd(z, d.gd_edge_index, mod="cat", if_relu=True)    # x, tagged with its bf16 split planes

    def _decode_eager(self):
        return self.model.dmt(self.z, self.idx, self.et)

    def genes(self):
        self.x = self._genes()
        return self.x

    def drugs(self):
        return self._drugs()

    def _drugs_eager(self):
        d = self.data
        self.z = self.model.dd(self.x, d.train_idx, edge_type=d.train_et, range_list=d.train_range, if_catout=True)
        return self.z

    def decode(self):
        return self._decode()

    def _step_eager(self):
        if self.model.split_relational and not torch.is_grad_enabled():
            self.conv.start_pair_sums(self.data.train_idx, self.data.train_range, int(self.data.n_d_node))
        self.x = self._genes_eager()
        self._drugs_eager()
        return self.z, self._decode_eager()

    def step(self):
        if self._whole is not None:
            return self._whole()
        if self._encode is not None:
            self._encode()
        else:
            self.genes()
            self.drugs()
        return self.z, self.decode()
