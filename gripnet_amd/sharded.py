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

import os
from typing import Optional

import torch

from .utils import shard_edge_ranges


class HipShardKernels:
    """The product arithmetic: HIP kernels behind the C ABI (no CPU fallback)."""

    def __init__(self, model, data, lo, hi):
        from . import _hip
        self._hip, self.model, self.data = _hip, model, data
        self.conv = model.dd.conv_list[0]
        self.plan = _hip.RgcnPlan(data.train_idx, data.train_range, data.n_d_node, lo, hi)
        self.idx = data.train_idx[:, lo:hi].contiguous()
        self.et = data.train_et[lo:hi].contiguous()

    def encode_genes(self):
        z = self.model.gg(None, self.data.gg_edge_index, edge_weight=self.data.edge_weight, if_catout=True)
        return self.model.gd(z, self.data.gd_edge_index, mod="cat", if_relu=True)

    def partial(self, x, out, fresh_weights=False):
        """(`fresh_weights` stays in the signature the CPU stand-in kernels of the gloo tests share: every call computes what it needs.)"""
        c = self.conv
        planes = self._hip.SplitPlanes.of(x, c.in_channels // 16) if c.in_channels % 16 == 0 else None
        return self.plan.forward(x, c.basis.detach(), c.att.detach(), None, None, False, out, partial=True,
                                 fast=c._fast(), path=c.kernel, x_planes=planes)

    def finalize(self, summed, x, out, slot0):
        c = self.conv                                           # concat slot 0 is copied by the same launch
        side = None if slot0 is None else (x, slot0, 0)
        return self.plan.finalize(summed, x, c.root.detach(), None if c.bias is None else c.bias.detach(), True, out, side=side)

    def parameters(self):
        return self.model.parameters()

    def decoder_weight(self):
        return self.model.dmt.weight

    def score(self, z, sigmoid=True):
        return self.model.dmt(z, self.idx, self.et, sigmoid=sigmoid)

    def score_input_columns(self, x, sigmoid=True):
        """Starts the scores of my edge range on the columns of z that are the relational layer's INPUT (concat slot 0,
        layers.py:264-266): they do not wait for the all-reduce.  Returns what score_rest needs, or None when the list
        has no static plan (then score() does everything after the exchange)."""
        dmt = self.model.dmt
        if torch.is_grad_enabled() and (x.requires_grad or dmt.weight.requires_grad):
            return None
        if x.shape[1] % 4 != 0 or x.shape[1] >= dmt.in_dim:
            return None
        dmt.register_static(self.idx, self.et, num_nodes=x.shape[0])     # my edge range is scored every step
        plan = dmt.plan_for(x, self.idx, self.et)
        if plan is None or plan.num_nodes != x.shape[0]:
            return None
        out = torch.empty((self.idx.shape[1],), dtype=torch.float32, device=x.device)
        try:
            plan.forward_cols(self._hip.f32_rows(x), dmt.in_dim, 0, x.shape[1], dmt.weight, sigmoid, out)
        except self._hip.GripNetHipError as err:                         # node table too large for the LDS
            if err.status != self._hip.GN_ERR_UNSUPPORTED:
                raise
            return None
        return plan, out, x.shape[1]

    def score_rest(self, started, z, sigmoid=True):
        plan, out, col = started
        return plan.forward_cols(z, z.shape[1], col, z.shape[1], self.model.dmt.weight, sigmoid, out)

    # ---- training (autograd is recording: the modules route through gripnet_amd.autograd) ----
    def edge_gradients(self, x, gm):
        """This shard's share of the sums over edges in the relational layer's gradient."""
        from .autograd import rgcn_edge_gradients
        c = self.conv
        return rgcn_edge_gradients(self.plan, x, c.basis.detach(), c.att.detach(), gm)

    def rgcn_parameters(self):
        c = self.conv
        return c.basis, c.att, c.root, c.bias

    def in_degree(self):
        return self.plan.grad_plans()[2]

    def score_edges(self, z, edge_index, sigmoid=True):
        """Scores of my edge range's relations on another edge list of the same length (negative samples)."""
        return self.model.dmt(z, edge_index, self.et, sigmoid=sigmoid)

    def layer_gradient(self, g, out, want_bias):
        """ReLU mask by the saved output, gm = g / in-degree and the bias gradient: one launch (gn_grad_prologue_f32)."""
        return self._hip.grad_prologue(g, out, self.in_degree(), True, bool(want_bias))

    def root_gradients(self, g, root, x, dx):
        """dx += g root^T and droot = x^T g on the library's own products."""
        dx = self._hip.gemm(g, root, dx.contiguous(), b_transposed=True, accumulate=True)
        return dx, self._hip.xtg(x, g)

    def shard_loss(self, pos, neg, total_edges, eps):
        """This rank's share of the loss of GripNet-pose.py:140-142 in one launch each way: link_loss is -mean(log pos) -
        mean(log(1 - neg)) over the shard's edges; times (shard edges / all edges) it is the shard's share of the means
        over ALL edges."""
        from .utils import link_loss
        if pos.numel() != neg.numel() or pos.numel() == 0:
            raise ValueError("a shard scores as many negatives as positives, and at least one ({} / {})".format(pos.numel(), neg.numel()))
        return link_loss(pos, neg, eps) * (pos.numel() / float(total_edges))


class OneShotAllReduce:
    """The exchange step as ONE hop over xGMI's point-to-point links instead of a collective-library call (82 KB
    are latency-bound; a ring serialises 2 (G - 1) hops): every rank writes its partial into its slot of every peer's
    buffer, raises a flag, waits for its own G flags ON THE DEVICE and adds the G slots in rank order
    (csrc/exchange.hip) - all ranks get the same bits, nothing synchronises the host.

    Setup is collective (every rank of `group` constructs it with the same `numel`): the slot buffers and flag arrays are
    shared as IPC handles (torch's CUDA-IPC storages, i.e. hipIpcGetMemHandle / hipIpcOpenMemHandle) gathered with
    `all_gather_object`; afterwards the exchange makes no torch.distributed call.  ``all_reduce(t)`` sums in place, on the
    current stream.  Cannot be timed on a one-GPU box; tested there with two and three PROCESSES sharing the GPU against the
    rank-order sum (tests/test_gpu_callers.py)."""

    def __init__(self, numel: int, device, rank: int, world_size: int, group=None, timeout_ms: int = 2000):
        import torch.distributed as dist
        from . import _hip
        self._hip, self.rank, self.world, self.numel = _hip, int(rank), int(world_size), int(numel)
        self.device, self.timeout_ms, self.step = torch.device(device), int(timeout_ms), 0
        lib = _hip.load()
        nbytes = int(lib.gn_exchange_buffer_bytes(self.numel, self.world))
        # (an IPC handle names the allocator's whole block and the tensor's offset inside it: torch's storages carry both)
        self.slots = torch.zeros((max(nbytes // 4, 1),), dtype=torch.float32, device=self.device)
        self.flags = torch.zeros((self.world,), dtype=torch.int32, device=self.device)
        self.ticket = torch.zeros((1,), dtype=torch.int32, device=self.device)
        torch.cuda.synchronize(self.device)
        mine = (self.slots.untyped_storage()._share_cuda_(), self.flags.untyped_storage()._share_cuda_())
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        self._peers = []                                             # (keeps the opened storages alive)
        slot_ptrs, flag_ptrs = [], []
        for r, (hs, hf) in enumerate(handles):
            if r == self.rank:
                ts, tf = self.slots, self.flags
            else:
                # (the opened storage lives on the EXPORTER's device - another GPU of the node; opening it enables peer access, so
                # this rank's launches may write through the pointer)
                ss, sf = torch.UntypedStorage._new_shared_cuda(*hs), torch.UntypedStorage._new_shared_cuda(*hf)
                ts = torch.empty(0, dtype=torch.float32, device=ss.device).set_(ss)
                tf = torch.empty(0, dtype=torch.int32, device=sf.device).set_(sf)
            self._peers.append((ts, tf))
            slot_ptrs.append(ts.data_ptr())
            flag_ptrs.append(tf.data_ptr())
        import ctypes
        self._slot_ptrs = (ctypes.c_void_p * self.world)(*slot_ptrs)
        self._flag_ptrs = (ctypes.c_void_p * self.world)(*flag_ptrs)
        dist.barrier(group=group)                                    # every rank has mapped every buffer before the first push

    def all_reduce(self, t: torch.Tensor):
        h = self._hip
        if t.numel() != self.numel or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("one-shot exchange: a contiguous fp32 tensor of {} elements, got {} of {}".format(self.numel, t.dtype, tuple(t.shape)))
        self.step += 1
        import ctypes
        st = h.stream_ptr(t.device)
        h._call("gn_exchange_push_f32", h.ptr(t), self.numel, ctypes.addressof(self._slot_ptrs), ctypes.addressof(self._flag_ptrs),
                h.ptr(self.ticket), self.world, self.rank, self.step, st)
        h._call("gn_exchange_wait_sum_f32", h.ptr(self.slots), h.ptr(self.flags), self.numel, self.world, self.step, h.ptr(t),
                self.timeout_ms, h.ptr(h.error_flag(t.device)), h.stream_ptr(t.device))
        return t


class ShardedPoseForward:
    """``z, score_of_my_edge_range = fwd()`` on every rank; ``z`` is identical on all ranks."""

    def __init__(self, model, data, rank: int, world_size: int, group=None, kernels=None):
        self.rank, self.world_size, self.group = int(rank), int(world_size), group
        self.n_d = int(data.n_d_node)
        E = int(data.train_idx.shape[1])
        self.total_edges = E
        self.edge_lo, self.edge_hi = shard_edge_ranges(E, self.world_size)[self.rank]
        self.kernels = kernels if kernels is not None else HipShardKernels(model, data, self.edge_lo, self.edge_hi)
        conv = model.dd.conv_list[0]
        self.in_dim, self.out_dim = conv.in_channels, conv.out_channels
        dev = data.train_idx.device
        self._partial = torch.empty((self.n_d, self.out_dim), dtype=torch.float32, device=dev)
        # None: score the input columns beside the exchange when there is one; True / False force it (tests, measurements)
        env = os.environ.get("GN_SHARD_OVERLAP")
        self.overlap_decoder = None if env is None else env == "1"
        # issue the collectives also at world_size 1 (they are the identity there): lets one GPU exercise the RCCL calls,
        # their stream and their event path (tests/test_gpu_callers.py)
        self.always_exchange = False
        self.one_shot = None                     # use_one_shot_exchange(): the forward's exchange as one direct hop

    def use_one_shot_exchange(self, timeout_ms: int = 2000):
        """Collective: from now on the forward's all-reduce of the partial [n_d, out] is the one-shot direct exchange
        (OneShotAllReduce) instead of `dist.all_reduce`.  (The training step's other exchanges stay on the collective library.)"""
        self.one_shot = OneShotAllReduce(self._partial.numel(), self._partial.device, self.rank, self.world_size, self.group, timeout_ms)
        return self

    def all_reduce(self, t: torch.Tensor):
        if self.one_shot is not None and t.data_ptr() == self._partial.data_ptr():
            return self.one_shot.all_reduce(t)
        if self.world_size > 1 or self.always_exchange:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_begin(self, t: torch.Tensor):
        """The exchange as an asynchronous collective: what is launched before `all_reduce_end` runs beside it."""
        if self.one_shot is not None and t.data_ptr() == self._partial.data_ptr():
            # stream-ordered launches on the current stream: the push goes out at once, the decoder's input-column launch that the
            # caller issues next is queued BEHIND the wait (one stream) - the overlap of the collective path needs a second stream
            self.one_shot.all_reduce(t)
            return None
        if self.world_size > 1 or self.always_exchange:
            import torch.distributed as dist
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    @staticmethod
    def all_reduce_end(work):
        if work is not None:
            work.wait()

    def __call__(self, sigmoid: bool = True):
        k = self.kernels
        x = k.encode_genes()                                          # [n_d, in_dim], replicated
        out = torch.empty((self.n_d, self.in_dim + self.out_dim), dtype=torch.float32, device=x.device)
        k.partial(x, self._partial)                                   # un-normalised sum over my edge range
        work = self.all_reduce_begin(self._partial)                   # the one exchange step of the path
        # While the 82 KB travel: the decoder's first column phase.  z = [x | layer output] and the first 48 of its 80
        # columns are x, which every rank already holds (a second launch: only worth it when there is an exchange).
        started = None
        if (self.overlap_decoder if self.overlap_decoder is not None else self.world_size > 1) and hasattr(k, "score_input_columns"):
            started = k.score_input_columns(x, sigmoid=sigmoid)
        self.all_reduce_end(work)
        # mean / root / bias / ReLU (layers.py:191-197,305) and concat slot 0 (layers.py:264-266)
        k.finalize(self._partial, x, out[:, self.in_dim:], out[:, :self.in_dim])
        if started is not None:
            return out, k.score_rest(started, out, sigmoid=sigmoid)
        return out, k.score(out, sigmoid=sigmoid)


    def record(self, sigmoid: bool = True):
        """The same forward with its launches recorded once and made again from one loop (gripnet_amd._hip.Recorder; see
        pipeline.Recorded): three lists of entry-point calls around the exchange - gene / external layers and my partial sums;
        the decoder's input columns (beside the all-reduce, when it overlaps); finalisation and the rest of the scores.  The
        collective itself is issued by torch.distributed in every call, as in __call__.  HIP kernels only."""
        from . import _hip
        k = self.kernels
        if not isinstance(k, HipShardKernels):
            raise TypeError("record() replays the library's entry points: it needs HipShardKernels")
        for _ in range(2):                                            # plans exist, my edge range is a registered static list
            self(sigmoid)
        torch.cuda.synchronize()
        overlap = (self.overlap_decoder if self.overlap_decoder is not None else self.world_size > 1) and hasattr(k, "score_input_columns")
        with _hip.Recorder() as first:
            x = k.encode_genes()
            k.partial(x, self._partial)
        out = torch.empty((self.n_d, self.in_dim + self.out_dim), dtype=torch.float32, device=x.device)
        started, beside = None, None
        if overlap:
            with _hip.Recorder() as beside:
                started = k.score_input_columns(x, sigmoid=sigmoid)
            if started is None:
                beside = None
        with _hip.Recorder() as last:
            k.finalize(self._partial, x, out[:, self.in_dim:], out[:, :self.in_dim])
            score = k.score_rest(started, out, sigmoid=sigmoid) if started is not None else k.score(out, sigmoid=sigmoid)
        torch.cuda.synchronize()

        def run():
            _hip.replay(first.calls)
            work = self.all_reduce_begin(self._partial)
            if beside is not None:
                _hip.replay(beside.calls)
            self.all_reduce_end(work)
            _hip.replay(last.calls)
            return out, score
        run._recordings = (first, beside, last, x)                    # (the recorders hold the operands of their calls)
        return run


# ---- training: the same sharding with gradients (the reference trains every epoch,
#      GripNet-pose.py:140-146) ------------------------------------------------------------------------------------
class _SumGradAcrossRanks(torch.autograd.Function):
    """Identity forward; backward all-reduces the gradient.  z is replicated, every rank scores its own edge range
    with it: d loss / d z is the sum of the ranks' contributions."""

    @staticmethod
    def forward(ctx, z, owner):
        ctx.owner = owner
        return z.view_as(z)

    @staticmethod
    def backward(ctx, g):
        return ctx.owner.all_reduce(g.contiguous()), None


class _ShardedRgcnFn(torch.autograd.Function):
    """out = relu( all_reduce(P_k) / deg + x root + b ), P_k = sum over my edge range of x[src] W_r.
    Backward: the all-reduced partial's gradient gm = g / deg is the same on every rank (the all-reduce of a sum is
    the identity on its gradient); each rank reduces its own edge range into (dx, dbasis, datt) shares, ONE all-reduce
    of the three adds them up; the root / bias terms are replicated arithmetic on the full gradient."""

    @staticmethod
    def forward(ctx, x, basis, att, root, bias, owner):
        k = owner.kernels
        xc = x.detach()
        partial = torch.empty((xc.shape[0], basis.shape[2]), dtype=torch.float32, device=xc.device)
        k.partial(xc, partial, fresh_weights=True)
        owner.all_reduce(partial)
        out = torch.empty_like(partial)
        k.finalize(partial, xc, out, None)
        ctx.owner = owner
        ctx.save_for_backward(xc, root.detach(), out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, root, out = ctx.saved_tensors
        owner = ctx.owner
        has_bias = ctx.needs_input_grad[4]
        g, gm, dbias = owner.kernels.layer_gradient(g, out, has_bias)      # ReLU mask, gm = g / deg, bias gradient
        dxe, dbasis, datt = owner.kernels.edge_gradients(x, gm)
        flat = torch.cat([dxe.reshape(-1), dbasis.reshape(-1), datt.reshape(-1)])
        owner.all_reduce(flat)                                             # the one exchange step of the backward
        a, b = dxe.numel(), dxe.numel() + dbasis.numel()
        dx, droot = owner.kernels.root_gradients(g, root, x, flat[:a].view_as(dxe))   # + g root^T, x^T g
        return (dx, flat[a:b].view_as(dbasis), flat[b:].view_as(datt), droot, dbias, None)


class ShardedPoseTraining(ShardedPoseForward):
    """One training step of the PoSE model with the dd edges sharded over the ranks:

        loss = step(neg_index)        # leaves .grad on every parameter, identical on all ranks

    Exchange steps: the forward's all-reduce of the partial [n_d, out]; in the backward one all-reduce of d loss / d z
    ([n_d, in + out]), one of the shards' (dx, dbasis, datt) shares, and one of the decoder weight's gradient - each
    rank scores only its own edge range.  The gene layers are replicated arithmetic on replicated inputs: their
    gradients come out identical on every rank and are NOT reduced.  Loss of GripNet-pose.py:140-142."""

    EPS = 1e-13

    def step(self, neg_index, zero_grad: bool = True):
        k = self.kernels
        params = list(self.parameters())
        if zero_grad:
            for p in params:
                p.grad = None
        x = k.encode_genes()                                               # replicated, autograd-tracked
        basis, att, root, bias = k.rgcn_parameters()
        out = _ShardedRgcnFn.apply(x, basis, att, root, bias, self)
        z = _SumGradAcrossRanks.apply(torch.cat([x, out], dim=1), self)
        pos = k.score(z)
        neg = k.score_edges(z, neg_index[:, self.edge_lo:self.edge_hi].contiguous())
        local = k.shard_loss(pos, neg, self.total_edges, self.EPS)
        dw = k.decoder_weight()
        kept, dw.grad = dw.grad, None                                      # only THIS step's share is exchanged: what
        local.backward()                                                   # earlier steps accumulated is already a sum
        if dw.grad is not None:                                            # over ranks (zero_grad=False)
            self.all_reduce(dw.grad)
            if kept is not None:
                dw.grad += kept
        else:
            dw.grad = kept
        loss = local.detach().clone()
        return self.all_reduce(loss)

    def parameters(self):
        return self.kernels.parameters()
