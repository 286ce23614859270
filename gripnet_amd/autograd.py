"""Backward pass of the supergraph propagation path.

The reference trains full-batch: every epoch calls ``loss.backward()`` through the layers
(GripNet-pose.py:140-146).  When autograd is recording, the modules in layers.py / decoder.py route
through the ``torch.autograd.Function``s below instead of the slot-fused inference path.  The sparse
parts of every gradient run in HIP kernels behind the C ABI (transposed normalised adjacency,
DistMult scatter), and so do the dense contractions: the tall-skinny weight gradients dW = x^T g (gn_xtg_f32), the relational
dW_r = X^T Q_r (gn_rel_weight_grad_f32), dx = g W^T and the basis / attention gradients (gn_gemm_f32 with operands given
transposed; x^T g wider than 64 x 32 outputs in tiles), the class decoder's row gather / scatter-add through a one-edge-per-node
plan, the merges of the external layer's mod="add" branches and of freebase-c, the relational weight gradient of graphs beyond
the fused kernel per relation on gn_rgcn_weight_grad_f32 (the all-nodes baseline of rgcn_pose.py).  torch arithmetic is left
for shapes outside every kernel only (a relational layer with more than 128 input or 64 output features: rgcn_edge_gradients'
index_add_ + batched matmul per slab of relations; more than 64 bases: att^T dW).
"""
from __future__ import annotations

import threading

import torch

from . import _hip


def recording(*tensors) -> bool:
    """True when autograd is on and at least one of the tensors takes part in it."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class Slot:
    """Columns [lo, hi) of a concat buffer (not a tensor argument: autograd sees the view a Function makes of it as that
    Function's own output)."""
    __slots__ = ("buf", "lo", "hi")

    def __init__(self, buf, lo, hi):
        self.buf, self.lo, self.hi = buf, lo, hi

    def view(self):
        return self.buf[:, self.lo:self.hi]


def cat_slots(widths, n_rows, device):
    buf = torch.empty((n_rows, int(sum(widths))), dtype=torch.float32, device=device)
    slots, lo = [], 0
    for w in widths:
        slots.append(Slot(buf, lo, lo + int(w)))
        lo += int(w)
    return buf, slots


class SlotsCatFn(torch.autograd.Function):
    """``torch.cat(parts, dim=1)`` (layers.py:309,376) for parts that already sit in their columns of one buffer - written
    there by the launches that produced them (GcnConvFn / RgcnConvFn with a slot, their side copies): no copy forward, column
    slices of the gradient backward."""

    @staticmethod
    def forward(ctx, slots, *parts):
        ctx.bounds = [(s.lo, s.hi) for s in slots]
        return slots[0].buf.view_as(slots[0].buf)

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g[:, lo:hi] if need else None for (lo, hi), need in zip(ctx.bounds, ctx.needs_input_grad[1:]))


class AbsSlotFn(torch.autograd.Function):
    """``torch.abs(t)`` (layers.py:376) whose value a side copy of another launch already left in `slot`."""

    @staticmethod
    def forward(ctx, t, slot):
        ctx.save_for_backward(t)
        return slot.view()

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        if g.dtype != torch.float32 or g.dim() != 2 or (g.shape[1] > 1 and g.stride(1) != 1) or not t.is_contiguous():
            return g * torch.sign(t), None
        out = torch.empty_like(t)
        _hip.merge(out, g, 6, t.detach())                      # g * sign(t) in one launch (g: a column slice of the concat's gradient)
        return out, None


class GcnConvFn(torch.autograd.Function):
    """``act(A_norm (x W) + b)`` over a cached plan (myGCN.forward, gripnet/layers.py:71-100, with the ReLU
    that follows it at layers.py:279,305,370).  ``n_out`` rows: N for a square graph, n_target for the
    external layer's closed form."""

    @staticmethod
    def forward(ctx, x, weight, bias, plan, n_out, relu, slot=None, side=None, planes=None, passthrough=False, storage="fp32"):
        # slot: a Slot of a concat buffer the output is written into (the concat of layers.py:309,376 without a copy, SlotsCatFn);
        # side: (tensor, Slot, mode) copied by the same launch, as on the inference path; planes: (SplitPlanes, col_main, col_side) -
        # the launch leaves its output and side copy as bf16 split planes too (the external layer in front of a relational layer);
        # passthrough: x is ALSO returned, for the concat that holds it next to this layer's output (layers.py:280-281,307-309):
        # x then has this Function as its only consumer, and backward adds the concat's gradient columns of x where it stores
        # dx (the addend of gn_gemm_addend_f32) - the autograd engine's own sum of two gradients was one launch more per layer
        x_in = x
        ctx.set_materialize_grads(False)                       # (an output nobody differentiates arrives as None, not as a fresh matrix of zeros)
        x = _hip.f32_rows(x.detach())
        w = weight.detach()
        out = slot.view() if slot is not None else torch.empty((n_out, w.shape[1]), dtype=torch.float32, device=x.device)
        if side is not None:
            side = (side[0].detach(), side[1].view(), side[2])
        b = None if bias is None else bias.detach()
        if (storage == "bf16" and planes is None and w.shape[1] % 8 == 0 and _hip.ld(out) % 4 == 0 and out.data_ptr() % 16 == 0):
            # bf16 storage of the gathered table under training (round 6; BASELINE config 5): the forward is the inference path's
            # (x W rounded to bf16 where the product stores it, fp32 sums); the backward below is the fp32 layer's - rounding
            # the table has the identity as its (straight-through) derivative, and neither dx = A^T g W^T nor dW = x^T A^T g
            # reads the table.  What differs from fp32 training is the forward's values (and the ReLU mask they give).
            xw = torch.empty((x.shape[0], w.shape[1]), dtype=torch.bfloat16, device=x.device)
            try:
                _hip.gemm(x, w, xw, out_bf16=True)
            except _hip.GripNetHipError as err:
                if err.status != _hip.GN_ERR_UNSUPPORTED:
                    raise
                xw = torch.empty((x.shape[0], w.shape[1]), dtype=torch.float32, device=x.device)
                _hip.gemm(x, w, xw)
            plan.aggregate_bf16(xw, b, relu, out, side)
        elif _hip.transform_fusable(w.shape[0], w.shape[1], x) and w.is_contiguous():
            plan.aggregate(x, b, relu, out, side, weight=w, planes=planes)    # (A_norm x) W in one launch, as the inference path
        else:
            xw = torch.empty((x.shape[0], w.shape[1]), dtype=torch.float32, device=x.device)
            _hip.gemm(x, w, xw)
            plan.aggregate(xw, b, relu, out, side, planes=planes)
        ctx.plan, ctx.relu, ctx.has_bias = plan, bool(relu), bias is not None
        ctx.save_for_backward(x, w, out if relu else None)
        return (out, x_in) if passthrough else out

    @staticmethod
    def backward(ctx, g, g_pass=None):
        x, w, out = ctx.saved_tensors
        if g is None:                                              # only the passed-through input is used downstream
            return (g_pass if ctx.needs_input_grad[0] else None,) + (None,) * 10
        # one launch: the ReLU mask by the saved output (gradient passes where the output is positive), the bias gradient
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.relu or need_db or g.stride(1) != 1:
            gm, _, db = _hip.grad_prologue(g, out if ctx.relu else None, None, True, need_db)
        else:
            gm, db = _hip.f32_rows(g), None
        gxw = torch.empty((ctx.plan.n_table, gm.shape[1]), dtype=torch.float32, device=gm.device)
        ctx.plan.aggregate_t(gm, gxw)                          # A_norm^T g  (HIP, source-major CSR)
        dx = None
        with _hip.dense_batch(gxw.device):                     # dx and dW do not depend on each other: one launch
            late = None
            if ctx.needs_input_grad[0]:                        # gxw W^T (gn_gemm_f32, W given as it is stored) + the concat's columns of x
                dx = torch.empty((gxw.shape[0], w.shape[0]), dtype=torch.float32, device=gxw.device)
                rides = _hip.addend_ok(g_pass, dx.shape[0], dx.shape[1])
                _hip.gemm(gxw, w, dx, b_transposed=True, join_batch=True, addend=g_pass if rides else None)
                late = None if rides else g_pass
            dw = _hip.xtg(x, gxw, join_batch=True) if ctx.needs_input_grad[1] else None
        if late is not None:
            dx = dx + late
        return dx, dw, db, None, None, None, None, None, None, None, None


class RgcnConvFn(torch.autograd.Function):
    """``act(mean_{e: dst=i} x[src_e] W_{r(e)} + x[i] root + b)`` (myRGCN.forward, layers.py:165-197)."""

    @staticmethod
    def forward(ctx, x, basis, att, root, bias, plan, relu, slot=None, side=None, x_planes=None, passthrough=False):
        # (slot, side, passthrough: as GcnConvFn's)
        ctx.set_materialize_grads(False)
        xc = _hip.f32_rows(x.detach())
        out = slot.view() if slot is not None else torch.empty((xc.shape[0], basis.shape[2]), dtype=torch.float32, device=xc.device)
        if side is not None:
            side = (side[0].detach(), side[1].view(), side[2])
        plan.forward(xc, basis.detach(), att.detach(), root.detach(), None if bias is None else bias.detach(), relu, out, side=side,
                     x_planes=x_planes)
        ctx.plan, ctx.relu = plan, bool(relu)
        ctx.save_for_backward(xc, basis, att, root, out if relu else None)
        return (out, x) if passthrough else out

    @staticmethod
    def backward(ctx, g, g_pass=None):
        # out[i] = (sum_{e: dst=i} x[src_e] W_{r(e)}) / deg_i + x[i] root + b,   W_r = sum_b att[r,b] basis[b]
        x, basis, att, root, out = ctx.saved_tensors
        if g is None:
            return (g_pass if ctx.needs_input_grad[0] else None,) + (None,) * 10
        deg = ctx.plan.grad_plans()[2]
        # one launch: ReLU mask, gm = g / deg (the gradient of the un-normalised sum), the bias gradient
        g, gm, dbias = _hip.grad_prologue(g, out if ctx.relu else None, deg, True, bool(ctx.needs_input_grad[4]))
        # dbasis, datt and droot are independent, deep and small: they leave as ONE launch at the end of the block
        with _hip.dense_batch(x.device):
            dxe, dbasis, datt = rgcn_edge_gradients(ctx.plan, x, basis.detach(), att.detach(), gm,
                                                    ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2])
            dx, late = None, None
            if ctx.needs_input_grad[0]:                        # dx = dxe + g root^T (+ the concat's columns of x): added onto the edge sums
                rides = _hip.addend_ok(g_pass, dxe.shape[0], dxe.shape[1])
                dx = _hip.gemm(g, root.detach(), dxe, b_transposed=True, accumulate=True, join_batch=True, addend=g_pass if rides else None)
                late = None if rides else g_pass
            droot = _hip.xtg(x, g, join_batch=True) if ctx.needs_input_grad[3] else None
        if late is not None:
            dx = dx + late
        return dx, dbasis, datt, droot, dbias, None, None, None, None, None, None


# (relation, source) sums Q of more than this many floats are reduced per slab of relations (the dense Q of a graph
# with 10^3 relations x 10^5 nodes would need over 12 GB where the reference's per-relation loop needs O(E))
Q_BUDGET_FLOATS = 1 << 26


def rgcn_edge_gradients(plan, x, basis, att, gm, need_x=True, need_basis=True, need_att=True):
    """The parts of the relational layer's gradient that are sums over the plan's edges (a shard's edge range gives
    that shard's share; the shares add up): for P = sum_e x[src_e] W_{r(e)} and gm = dL/dP,
        dx[s]  = sum_{e: src=s} gm[dst_e] W_{r(e)}^T
        dW_r   = X^T Q_r,  Q_r[s] = sum_{e in r, src=s} gm[dst_e];   dbasis = att^T dW,  datt = dW basis^T."""
    n, fin = x.shape
    B, _, fout = basis.shape
    R = att.shape[0]
    rev, pairs, _ = plan.grad_plans()
    dxe = dbasis = datt = None
    if need_x:
        # the same relational layer on the reversed graph with the transposed bases, un-normalised (one launch of the
        # destination-major kernel where it applies: any output width that is a multiple of 4 up to 64)
        dxe = torch.empty((n, fin), dtype=torch.float32, device=x.device)
        bt = basis.transpose(1, 2)                                           # [B, fout, fin]: W_r^T = sum_b att[r,b] bt[b]
        if rev.path(fout, fin, B) != "pair" and fout == 32 and fin % 32 != 0 and fin > 32:
            # the LDS-resident kernel produces 32 output features: run it per 32-column block of W_r^T
            # (the last block zero-padded) instead of falling back to the HBM-table path
            for c0 in range(0, fin, 32):
                w = min(32, fin - c0)
                if w == 32:                                    # a full block: straight into its columns of dx
                    rev.forward(gm, bt[:, :, c0:c0 + 32].contiguous(), att, None, None, False, dxe[:, c0:c0 + 32], partial=True)
                    continue
                blk = torch.zeros((B, fout, 32), dtype=torch.float32, device=x.device)
                blk[:, :, :w] = bt[:, :, c0:c0 + w]
                tmp = torch.empty((n, 32), dtype=torch.float32, device=x.device)
                rev.forward(gm, blk, att, None, None, False, tmp, partial=True)
                dxe[:, c0:c0 + w] = tmp[:, :w]
        elif rev.path(fout, fin, B) == "pair" and basis.is_contiguous() and basis.data_ptr() % 16 == 0:
            # (the destination-major kernel reads W_r^T out of the forward's own parameter: no transposed copy per step)
            rev.forward(gm, basis, att, None, None, False, dxe, partial=True, basis_transposed=True)
        else:
            rev.forward(gm, bt.contiguous(), att, None, None, False, dxe, partial=True)
    if need_basis or need_att:
        # (HIP gather-reduce over (relation, source) rows, then library GEMMs; relation slabs when Q would be huge)
        wg = plan.weight_grad_plan()
        if wg is not None and x.stride(1) == 1 and gm.stride(1) == 1 and wg.supported(fin, fout):
            dw = wg.weight_grad(x, gm)                                       # X^T Q_r in one launch, Q never in memory
        elif x.stride(1) == 1 and gm.stride(1) == 1 and (dw_general := plan.general_weight_grad(x, gm)) is not None:
            dw = dw_general                                                  # per relation x[src]^T gm[dst], O(E) memory, any size
        elif R * n * fout <= Q_BUDGET_FLOATS:
            q = torch.empty((R * n, fout), dtype=torch.float32, device=x.device)
            pairs.aggregate(gm, None, False, q)
            dw = torch.matmul(x.t(), q.view(R, n, fout)).reshape(R, fin * fout)            # [R, fin*fout]
        else:
            dw = _relation_slab_dw(plan, x, gm, R, n, fin, fout)
        # dbasis = att^T dW, datt = dW basis^T (layers.py:172-173): deep, narrow products (gn_gemm_f32, operands as they are stored)
        if need_basis:
            if B <= 64:
                dbasis = torch.empty((B, fin * fout), dtype=torch.float32, device=x.device)
                _hip.gemm(att.contiguous(), dw, dbasis, a_transposed=True, join_batch=True)
                dbasis = dbasis.view(B, fin, fout)
            else:
                dbasis = (att.t() @ dw).view(B, fin, fout)
        if need_att:
            datt = torch.empty((R, B), dtype=torch.float32, device=x.device)
            _hip.gemm(dw, basis.reshape(B, fin * fout), datt, b_transposed=True, join_batch=True)
    return dxe, dbasis, datt


def _relation_slab_dw(plan, x, gm, R, n, fin, fout):
    """dW_r = X^T Q_r without the dense [R n, fout] Q: relations in slabs whose Q fits the budget, each slab's
    (relation, source) sums scattered from its edge range (O(edges) memory, like the reference's per-relation loop)."""
    ei, rl = plan._edge_index, plan._range_list
    dw = torch.zeros((R, fin * fout), dtype=torch.float32, device=x.device)
    slab = max(1, Q_BUDGET_FLOATS // max(1, n * fout))
    for r0 in range(0, R, slab):
        r1 = min(R, r0 + slab)
        lo, hi = max(int(rl[r0, 0]), plan.edge_lo), min(int(rl[r1 - 1, 1]), plan.edge_hi)
        if lo >= hi:
            continue
        sizes = (rl[r0:r1, 1].clamp(plan.edge_lo, plan.edge_hi) - rl[r0:r1, 0].clamp(plan.edge_lo, plan.edge_hi)).to(x.device)
        rel = torch.repeat_interleave(torch.arange(r1 - r0, device=x.device), sizes)
        q = torch.zeros(((r1 - r0) * n, fout), dtype=torch.float32, device=x.device)
        q.index_add_(0, rel * n + ei[0, lo:hi], gm.index_select(0, ei[1, lo:hi]))
        dw[r0:r1] = torch.matmul(x.t(), q.view(r1 - r0, n, fout)).reshape(r1 - r0, fin * fout)
    return dw


class DistMultFn(torch.autograd.Function):
    """``sigma?(sum_k z[u,k] z[v,k] D[r,k])`` (decoder.py:19-23) with the scatter gradients in HIP."""

    @staticmethod
    def forward(ctx, z, weight, edge_index, edge_type, sigmoid, plan=None):
        zc = _hip.f32_rows(z.detach())
        w = weight.detach()
        out = torch.empty((edge_index.shape[1],), dtype=torch.float32, device=zc.device)
        done = False
        if plan is not None:                                   # a static edge list (the positives): same bits, fewer bytes
            try:
                plan.forward(zc, w, sigmoid, out)
                done = True
            except _hip.GripNetHipError as err:                # node table too large for the LDS: the general kernels
                if err.status != _hip.GN_ERR_UNSUPPORTED:
                    raise
        if not done:
            _hip.distmult_any(zc, edge_index, edge_type, w, sigmoid, out)
        ctx.sigmoid = bool(sigmoid)
        ctx.plan = plan                                        # a static list: its backward plan hangs on the forward one
        ctx.save_for_backward(zc, w, edge_index, edge_type, out if sigmoid else None)
        return out

    @staticmethod
    def backward(ctx, g):
        z, w, ei, et, out = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        dz = torch.empty_like(z)
        dd = torch.empty_like(w)
        # d sigma(s) / d s = p (1 - p) is applied inside, where the edge records are built
        probs = out if ctx.sigmoid else None
        bwd = None
        if ctx.plan is not None:                               # sort once per static edge list, not once per step
            bwd = getattr(ctx.plan, "bwd", None)
            if bwd is None:
                try:
                    bwd = _hip.DistMultBwdPlan(ei, et, z.shape[0], w.shape[0])
                except _hip.GripNetHipError as err:            # tables too large for the LDS path
                    if err.status != _hip.GN_ERR_UNSUPPORTED:
                        raise
                    bwd = False
                ctx.plan.bwd = bwd
        done = False
        if bwd:
            try:
                bwd.backward(z, w, g, dz, dd, probs)
                done = True
            except _hip.GripNetHipError as err:                # unaligned rows: the general entry point
                if err.status != _hip.GN_ERR_UNSUPPORTED:
                    raise
        if not done:
            _hip.distmult_backward(z, ei, et, w, g, dz, dd, probs=probs)
        return (dz if ctx.needs_input_grad[0] else None), (dd if ctx.needs_input_grad[1] else None), None, None, None, None


_loss_ws = {}


def _loss_workspace(device):
    """Per-device scratch of gn_link_loss_forward_f32, zeroed once (the kernel leaves it ready for the next launch)."""
    key = (device.type, device.index)
    if key not in _loss_ws:
        _loss_ws[key] = torch.zeros((int(_hip.load().gn_link_loss_workspace_bytes()),), dtype=torch.uint8, device=device)
    return _loss_ws[key]


class LinkLossFn(torch.autograd.Function):
    """``-mean(log(pos + eps)) - mean(log(1 - neg + eps))`` (the loss of GripNet-pose.py:140-142) in one launch forward and
    one backward, instead of the ~20 element-wise and reduction launches of the torch expression."""

    @staticmethod
    def forward(ctx, pos, neg, eps):
        p, n = pos.detach().contiguous().float(), neg.detach().contiguous().float()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        ws = _loss_workspace(p.device)
        _hip._call("gn_link_loss_forward_f32", _hip.ptr(p), p.numel(), _hip.ptr(n), n.numel(), float(eps), _hip.ptr(loss),
                   _hip.ptr(ws), ws.numel(), _hip.stream_ptr(p.device))
        ctx.eps = float(eps)
        ctx.save_for_backward(p, n)
        return loss

    @staticmethod
    def backward(ctx, g):
        p, n = ctx.saved_tensors
        g = g.contiguous().float()
        dp, dn = torch.empty_like(p), torch.empty_like(n)
        _hip._call("gn_link_loss_backward_f32", _hip.ptr(p), p.numel(), _hip.ptr(n), n.numel(), ctx.eps, _hip.ptr(g), _hip.ptr(dp),
                   _hip.ptr(dn), _hip.stream_ptr(p.device))
        return (dp if ctx.needs_input_grad[0] else None), (dn if ctx.needs_input_grad[1] else None), None


class LinkPredictionLossFn(torch.autograd.Function):
    """``loss, pos_score, neg_score = f(z, D, pos_index, neg_index, edge_type)``: the two decoder calls of a training step and
    the loss over them (GripNet-pose.py:137-142) as ONE autograd node.  Forward: the same launches as the three separate calls
    (same bits).  Backward: the loss's derivative is a function of an edge's own probability, so each decoder backward computes
    it where it builds its edge records (gn_distmult_backward_loss_*_f32, csrc/distmult_bwd.hip: GradSrc) - no loss-backward
    launch, no [E] gradient vectors through memory, and dz / dD of the two lists are added here instead of by two engine
    launches on strided views.  The scores are returned for the epoch's metrics (GripNet-pose.py:148-160) and carry no
    gradient.  Where a fused launch does not apply (no static plan for the positives, no packed pairs for the negatives) that
    list takes the two-step path."""

    @staticmethod
    def forward(ctx, z, weight, pos_index, neg_index, edge_type, eps, plan):
        zc, w = _hip.f32_rows(z.detach()), weight.detach()
        E = pos_index.shape[1]
        pos = torch.empty((E,), dtype=torch.float32, device=zc.device)
        neg = torch.empty((neg_index.shape[1],), dtype=torch.float32, device=zc.device)
        done = False
        if plan is not None:
            try:
                plan.forward(zc, w, True, pos)
                done = True
            except _hip.GripNetHipError as err:
                if err.status != _hip.GN_ERR_UNSUPPORTED:
                    raise
        if not done:
            _hip.distmult_any(zc, pos_index, edge_type, w, True, pos)
        _hip.distmult_any(zc, neg_index, edge_type, w, True, neg)
        loss = torch.empty((), dtype=torch.float32, device=zc.device)
        ws = _loss_workspace(zc.device)
        _hip._call("gn_link_loss_forward_f32", _hip.ptr(pos), pos.numel(), _hip.ptr(neg), neg.numel(), float(eps), _hip.ptr(loss),
                   _hip.ptr(ws), ws.numel(), _hip.stream_ptr(zc.device))
        ctx.eps, ctx.plan = float(eps), plan
        ctx.save_for_backward(zc, w, pos_index, neg_index, edge_type, pos, neg)
        ctx.mark_non_differentiable(pos, neg)
        ctx.set_materialize_grads(False)                       # (the scores carry no gradient: no [E] zeros filled for them per step)
        return loss, pos, neg

    @staticmethod
    def backward(ctx, g, _gp, _gn):
        z, w, pos_index, neg_index, et, pos, neg = ctx.saved_tensors
        if g is None:
            return None, None, None, None, None, None, None
        g = g.contiguous().float()
        dz, dd = torch.empty_like(z), torch.empty_like(w)
        # ---- the positives: a static list with a backward plan -> the loss-fed planned launch ----
        bwd = None
        if ctx.plan is not None:
            bwd = getattr(ctx.plan, "bwd", None)
            if bwd is None:
                try:
                    bwd = _hip.DistMultBwdPlan(pos_index, et, z.shape[0], w.shape[0])
                except _hip.GripNetHipError as err:
                    if err.status != _hip.GN_ERR_UNSUPPORTED:
                        raise
                    bwd = False
                ctx.plan.bwd = bwd
        done = False
        if bwd:
            try:
                bwd.backward(z, w, None, dz, dd, pos, loss=_hip.LinkLossGrad(g, ctx.eps, False))
                done = True
            except _hip.GripNetHipError as err:
                if err.status != _hip.GN_ERR_UNSUPPORTED:
                    raise
        dn = None
        if not done:                                           # the two-step path: the loss's own backward, then the decoder's
            dp, dn = torch.empty_like(pos), torch.empty_like(neg)
            _hip._call("gn_link_loss_backward_f32", _hip.ptr(pos), pos.numel(), _hip.ptr(neg), neg.numel(), ctx.eps, _hip.ptr(g), _hip.ptr(dp),
                       _hip.ptr(dn), _hip.stream_ptr(z.device))
            _hip.distmult_backward(z, pos_index, et, w, dp, dz, dd, probs=pos)
        # ---- the negatives: the sampler's packed pairs -> the loss-fed packed launch ----
        # (the positives' gradients ride on its combine launch as addends: dz / dD of the two lists need no adding launch)
        if _hip.distmult_backward_loss_packed(z, neg_index, et, w, neg, dz, dd, _hip.LinkLossGrad(g, ctx.eps, True), dz_add=dz, dd_add=dd):
            return (dz if ctx.needs_input_grad[0] else None), (dd if ctx.needs_input_grad[1] else None), None, None, None, None, None
        dz2, dd2 = torch.empty_like(z), torch.empty_like(w)
        if not _hip.distmult_backward_loss_packed(z, neg_index, et, w, neg, dz2, dd2, _hip.LinkLossGrad(g, ctx.eps, True)):
            if dn is None:
                dp, dn = torch.empty_like(pos), torch.empty_like(neg)
                _hip._call("gn_link_loss_backward_f32", _hip.ptr(pos), pos.numel(), _hip.ptr(neg), neg.numel(), ctx.eps, _hip.ptr(g), _hip.ptr(dp),
                           _hip.ptr(dn), _hip.stream_ptr(z.device))
            _hip.distmult_backward(z, neg_index, et, w, dn, dz2, dd2, probs=neg)
        dz.add_(dz2)
        dd.add_(dd2)
        return (dz if ctx.needs_input_grad[0] else None), (dd if ctx.needs_input_grad[1] else None), None, None, None, None, None


_node_plans = []            # [node_list tensor of the caller, _version, rows of z, plan or None]: the row-gather plans of the last few node lists
_node_plans_lock = threading.Lock()
_node_plan_misses = 0       # lists in a row that were not in the cache


def node_gather_plan(key: torch.Tensor, nodes: torch.Tensor, num_rows: int):
    """Plain-sum plan with ONE edge per listed node (table row nodes[i] -> output row i): its aggregation is the row gather
    z[node_list] (decoder.py:42), its transposed aggregation the scatter-add of the gathered rows' gradients.  Built once
    per node list (the label splits of a training loop are static, GripNet-aminer.py:124-147) and remembered under the
    CALLER's tensor `key` (identity + version: an int32 or strided list is converted on every forward, the caller's object is
    what repeats; a data pointer would not do - the allocator hands a freed list's address to the next one).
    Returns None when the list should not get a plan now: a caller that makes a new list every step (two misses in a row: a
    build synchronises the stream and walks the list on the host - every step) or a stream that is being captured (a build
    cannot be captured); the caller then gathers and scatters without a plan, and a list that does come back gets its plan
    at its second sighting."""
    global _node_plan_misses
    capturing = torch.cuda.is_current_stream_capturing()
    with _node_plans_lock:                                     # (autograd runs a device's backward on its own thread)
        entry = None
        for e in _node_plans:
            if e[0] is key and e[1] == key._version and e[2] == num_rows:
                entry = e
                break
        if entry is not None:
            _node_plan_misses = 0
            if entry[3] is not None or capturing:
                return entry[3]
        else:
            _node_plan_misses += 1
            if _node_plan_misses > 2 or capturing:
                _node_plans.append([key, key._version, num_rows, None])
                del _node_plans[:-4]
                return None
    ei = torch.stack([nodes, torch.arange(nodes.shape[0], dtype=torch.int64, device=nodes.device)])
    plan = _hip.GraphPlan.plain_sum(ei, num_rows, nodes.shape[0])
    with _node_plans_lock:
        if entry is not None:
            entry[3] = plan
        else:
            _node_plans.append([key, key._version, num_rows, plan])
            del _node_plans[:-4]
    return plan


class ClassLossFn(torch.autograd.Function):
    """``-log(score[range(n), classes] + eps).mean()`` (the loss of GripNet-aminer.py:133) in one launch each way."""

    @staticmethod
    def forward(ctx, score, classes, eps):
        sc = _hip.f32_rows(score.detach())
        cl = _hip.i64_vec(classes)
        if cl.numel() != sc.shape[0]:
            raise ValueError("{} class ids for {} scored nodes".format(cl.numel(), sc.shape[0]))
        loss = torch.empty((), dtype=torch.float32, device=sc.device)
        _hip._call("gn_class_loss_forward_f32", _hip.ptr(sc), _hip.ld(sc), _hip.ptr(cl), sc.shape[0], sc.shape[1], float(eps), _hip.ptr(loss),
                   _hip.ptr(_hip.error_flag(sc.device)), _hip.stream_ptr(sc.device))
        ctx.eps = float(eps)
        ctx.save_for_backward(sc, cl)
        return loss

    @staticmethod
    def backward(ctx, g):
        sc, cl = ctx.saved_tensors
        g = g.contiguous().float()
        d = torch.empty(sc.shape, dtype=torch.float32, device=sc.device)
        _hip._call("gn_class_loss_backward_f32", _hip.ptr(sc), _hip.ld(sc), _hip.ptr(cl), sc.shape[0], sc.shape[1], ctx.eps, _hip.ptr(g),
                   _hip.ptr(d), _hip.ld(d), _hip.stream_ptr(sc.device))
        return d, None, None


class ClassLogitsFn(torch.autograd.Function):
    """``z[node_list] @ W`` (decoder.py:42); forward on gn_class_scores_f32, backward on the library's own kernels: the
    gathered rows and the scatter-add of their gradients through a one-edge-per-node plan (gn_graph_aggregate_f32 /
    gn_graph_aggregate_t_f32), the two products on gn_gemm_f32 / gn_xtg_f32."""

    @staticmethod
    def forward(ctx, z, weight, node_list):
        zc = _hip.f32_rows(z.detach())
        w = weight.detach()
        nodes = _hip.i64_vec(node_list)
        out = torch.empty((nodes.shape[0], w.shape[1]), dtype=torch.float32, device=zc.device)
        _hip.class_scores(zc, w, nodes, out, False)
        ctx.nodes, ctx.key = nodes, node_list                      # (the caller's tensor keys the gather plan's cache)
        ctx.save_for_backward(zc, w)
        return out

    @staticmethod
    def backward(ctx, g):
        z, w = ctx.saved_tensors
        nodes = ctx.nodes
        dz = dw = None
        g = _hip.f32_rows(g.contiguous())
        if nodes.shape[0] == 0:
            return (torch.zeros_like(z) if ctx.needs_input_grad[0] else None), (torch.zeros_like(w) if ctx.needs_input_grad[1] else None), None
        plan = node_gather_plan(ctx.key, nodes, z.shape[0])    # (None: a list that is new every step, or a captured stream - no plan)
        zsel = None
        if ctx.needs_input_grad[1]:                            # z[nodes]: a one-edge-per-row aggregation
            if plan is None:
                zsel = z.index_select(0, nodes)
            else:
                zsel = torch.empty((nodes.shape[0], z.shape[1]), dtype=torch.float32, device=z.device)
                plan.aggregate(z, None, False, zsel)
        gw = None
        with _hip.dense_batch(g.device):                       # the two products do not depend on each other: one launch
            if ctx.needs_input_grad[0]:                        # rows of g W^T (gn_gemm_f32, W as it is stored) ...
                gw = torch.empty((g.shape[0], w.shape[0]), dtype=torch.float32, device=g.device)
                _hip.gemm(g, w, gw, b_transposed=True, join_batch=True)
            if zsel is not None:                               # z[nodes]^T g (gn_xtg_f32)
                dw = _hip.xtg(zsel, g, join_batch=True)
        if gw is not None:                                     # ... added at the listed nodes (rows named twice add up)
            if plan is None:
                dz = torch.zeros_like(z).index_add_(0, nodes, gw)
            else:
                dz = torch.empty_like(z)
                plan.aggregate_t(gw, dz)
        return dz, dw, None


class SoftmaxRowsFn(torch.autograd.Function):
    """``torch.softmax(logits, dim=1)`` (decoder.py:43) on gn_softmax_rows_f32 / gn_softmax_rows_backward_f32."""

    @staticmethod
    def forward(ctx, logits):
        p = torch.empty(logits.shape, dtype=torch.float32, device=logits.device)
        _hip.merge(p, _hip.f32_rows(logits.detach()), 0)
        _hip.softmax_rows(p)
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        g = _hip.f32_rows(g)
        dx = torch.empty(p.shape, dtype=torch.float32, device=p.device)
        _hip._call("gn_softmax_rows_backward_f32", _hip.ptr(p), _hip.ld(p), _hip.ptr(g), _hip.ld(g), _hip.ptr(dx), _hip.ld(dx),
                   p.shape[0], p.shape[1], _hip.stream_ptr(p.device))
        return dx


class MergeMeanFn(torch.autograd.Function):
    """``(a + b + c) / 3`` (the three-way merge of GripNet-freebase-c.py:158-162) in two launches forward (a slot copy and
    gn_merge_f32 mode 4) and one backward (every operand's gradient is g / 3: mode 7)."""

    @staticmethod
    def forward(ctx, a, b, c):
        out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
        _hip.merge(out, _hip.f32_rows(a.detach()), 0)
        _hip.merge(out, _hip.f32_rows(b.detach()), 4, src2=_hip.f32_rows(c.detach()))
        return out

    @staticmethod
    def backward(ctx, g):
        share = torch.empty(g.shape, dtype=torch.float32, device=g.device)
        _hip.merge(share, _hip.f32_rows(g), 7)
        return tuple(share if need else None for need in ctx.needs_input_grad)


class HalfSumAbsFn(torch.autograd.Function):
    """``(y + |t|) / 2`` (interGraph with mod != "cat" and equal widths, layers.py:378-379)."""

    @staticmethod
    def forward(ctx, y, t):
        tc = _hip.f32_rows(t.detach())
        out = torch.empty(y.shape, dtype=torch.float32, device=y.device)
        _hip.merge(out, _hip.f32_rows(y.detach()), 0)
        _hip.merge(out, tc, 2)
        ctx.save_for_backward(tc)
        return out

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        g = _hip.f32_rows(g)
        gy = gt = None
        if ctx.needs_input_grad[0]:
            gy = torch.empty(g.shape, dtype=torch.float32, device=g.device)
            _hip.merge(gy, g, 8)
        if ctx.needs_input_grad[1]:
            gt = torch.empty(g.shape, dtype=torch.float32, device=g.device)
            _hip.merge(gt, g, 9, t)
        return gy, gt


class HalfSumDownFn(torch.autograd.Function):
    """``(y + relu(t @ down)) / 2`` (interGraph with mod != "cat" and unequal widths, layers.py:381-384): the product on
    gn_gemm_f32, its gradients on gn_gemm_f32 / gn_xtg_f32."""

    @staticmethod
    def forward(ctx, y, t, down):
        tc, dc = _hip.f32_rows(t.detach()), _hip.f32_rows(down.detach())
        proj = torch.empty((tc.shape[0], dc.shape[1]), dtype=torch.float32, device=tc.device)
        _hip.gemm(tc, dc, proj)
        out = torch.empty(y.shape, dtype=torch.float32, device=y.device)
        _hip.merge(out, _hip.f32_rows(y.detach()), 0)
        _hip.merge(out, proj, 3)
        ctx.save_for_backward(tc, dc, proj)
        return out

    @staticmethod
    def backward(ctx, g):
        t, down, proj = ctx.saved_tensors
        g = _hip.f32_rows(g)
        gy = gt = gd = None
        if ctx.needs_input_grad[0]:
            gy = torch.empty(g.shape, dtype=torch.float32, device=g.device)
            _hip.merge(gy, g, 8)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gp = torch.empty(g.shape, dtype=torch.float32, device=g.device)
            _hip.merge(gp, g, 10, proj)                            # g / 2 where the projection was positive
            if ctx.needs_input_grad[1]:
                gt = torch.empty(t.shape, dtype=torch.float32, device=g.device)
                _hip.gemm(gp, down, gt, b_transposed=True)
            if ctx.needs_input_grad[2]:
                gd = _hip.xtg(t, gp)
        return gy, gt, gd
