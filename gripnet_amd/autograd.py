"""Backward pass of the supergraph propagation path.

The reference trains full-batch: every epoch calls ``loss.backward()`` through the layers
(GripNet-pose.py:140-146).  When autograd is recording, the modules in layers.py / decoder.py route
through the ``torch.autograd.Function``s below instead of the slot-fused inference path.  The sparse
parts of every gradient run in HIP kernels behind the C ABI (transposed normalised adjacency,
DistMult scatter) and so do the tall-skinny weight gradients dW = x^T g (gn_xtg_f32); the remaining small dense
contractions (dx = g W^T, the basis / attention gradients) are plain library GEMMs (torch.matmul).
"""
from __future__ import annotations

import torch

from . import _hip


def recording(*tensors) -> bool:
    """True when autograd is on and at least one of the tensors takes part in it."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class GcnConvFn(torch.autograd.Function):
    """``act(A_norm (x W) + b)`` over a cached plan (myGCN.forward, gripnet/layers.py:71-100, with the ReLU
    that follows it at layers.py:279,305,370).  ``n_out`` rows: N for a square graph, n_target for the
    external layer's closed form."""

    @staticmethod
    def forward(ctx, x, weight, bias, plan, n_out, relu):
        x = _hip.f32_rows(x.detach())
        w = weight.detach()
        out = torch.empty((n_out, w.shape[1]), dtype=torch.float32, device=x.device)
        b = None if bias is None else bias.detach()
        if _hip.transform_fusable(w.shape[0], w.shape[1], x) and w.is_contiguous():
            plan.aggregate(x, b, relu, out, weight=w)          # (A_norm x) W in one launch, as the inference path
        else:
            xw = torch.empty((x.shape[0], w.shape[1]), dtype=torch.float32, device=x.device)
            _hip.gemm(x, w, xw)
            plan.aggregate(xw, b, relu, out)
        ctx.plan, ctx.relu, ctx.has_bias = plan, bool(relu), bias is not None
        ctx.save_for_backward(x, w, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, out = ctx.saved_tensors
        g = _hip.f32_rows(g.contiguous())
        if ctx.relu:                                           # gradient passes where the output is positive
            g = _hip.merge(torch.empty_like(g), g, 5, src2=out)
        gxw = torch.empty((ctx.plan.n_table, g.shape[1]), dtype=torch.float32, device=g.device)
        ctx.plan.aggregate_t(g, gxw)                           # A_norm^T g  (HIP, source-major CSR)
        dx = gxw @ w.t() if ctx.needs_input_grad[0] else None
        dw = _hip.xtg(x, gxw) if ctx.needs_input_grad[1] else None
        db = g.sum(dim=0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None


class RgcnConvFn(torch.autograd.Function):
    """``act(mean_{e: dst=i} x[src_e] W_{r(e)} + x[i] root + b)`` (myRGCN.forward, layers.py:165-197)."""

    @staticmethod
    def forward(ctx, x, basis, att, root, bias, plan, relu):
        xc = _hip.f32_rows(x.detach())
        out = torch.empty((xc.shape[0], basis.shape[2]), dtype=torch.float32, device=xc.device)
        plan.forward(xc, basis.detach(), att.detach(), root.detach(), None if bias is None else bias.detach(), relu, out)
        ctx.plan, ctx.relu = plan, bool(relu)
        ctx.save_for_backward(xc, basis, att, root, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        # out[i] = (sum_{e: dst=i} x[src_e] W_{r(e)}) / deg_i + x[i] root + b,   W_r = sum_b att[r,b] basis[b]
        x, basis, att, root, out = ctx.saved_tensors
        g = _hip.f32_rows(g.contiguous())
        if ctx.relu:
            g = _hip.merge(torch.empty_like(g), g, 5, src2=out)
        n, fin = x.shape
        B, _, fout = basis.shape
        R = att.shape[0]
        rev, pairs, deg = ctx.plan.grad_plans()
        gm = g / deg.view(-1, 1)                                            # gradient of the un-normalised sum
        dx = dbasis = datt = droot = dbias = None
        if ctx.needs_input_grad[0]:
            # dx[s] = sum_{e: src=s} gm[dst_e] W_{r(e)}^T: the same relational layer on the reversed graph
            # with the transposed bases, un-normalised (HIP, general path), plus the root term
            dxe = torch.empty((n, fin), dtype=torch.float32, device=x.device)
            bt = basis.detach().transpose(1, 2)                                  # [B, fout, fin]: W_r^T = sum_b att[r,b] bt[b]
            if fout == 32 and fin % 32 != 0 and fin > 32:
                # the LDS-resident kernel produces 32 output features: run it per 32-column block of W_r^T
                # (the last block zero-padded) instead of falling back to the HBM-table path
                for c0 in range(0, fin, 32):
                    w = min(32, fin - c0)
                    if w == 32:                                # a full block: straight into its columns of dx
                        rev.forward(gm, bt[:, :, c0:c0 + 32].contiguous(), att.detach(), None, None, False,
                                    dxe[:, c0:c0 + 32], partial=True)
                        continue
                    blk = torch.zeros((B, fout, 32), dtype=torch.float32, device=x.device)
                    blk[:, :, :w] = bt[:, :, c0:c0 + w]
                    tmp = torch.empty((n, 32), dtype=torch.float32, device=x.device)
                    rev.forward(gm, blk, att.detach(), None, None, False, tmp, partial=True)
                    dxe[:, c0:c0 + w] = tmp[:, :w]
            else:
                rev.forward(gm, bt.contiguous(), att.detach(), None, None, False, dxe, partial=True)
            dx = dxe + g @ root.detach().t()
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # dW_r = X^T Q_r,  Q_r[s] = sum_{e in r, src=s} gm[dst_e]   (HIP gather-reduce, (relation, source) rows)
            q = torch.empty((R * n, fout), dtype=torch.float32, device=x.device)
            pairs.aggregate(gm, None, False, q)
            dw = torch.matmul(x.t(), q.view(R, n, fout)).reshape(R, fin * fout)            # [R, fin*fout]
            if ctx.needs_input_grad[1]:
                dbasis = (att.detach().t() @ dw).view(B, fin, fout)
            if ctx.needs_input_grad[2]:
                datt = dw @ basis.detach().reshape(B, fin * fout).t()
        if ctx.needs_input_grad[3]:
            droot = _hip.xtg(x, g)
        if ctx.needs_input_grad[4]:
            dbias = g.sum(dim=0)
        return dx, dbasis, datt, droot, dbias, None, None


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
            _hip.distmult(zc, edge_index, edge_type, w, sigmoid, out)
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


class ClassLogitsFn(torch.autograd.Function):
    """``z[node_list] @ W`` (decoder.py:42); forward on the MFMA row-gather GEMM."""

    @staticmethod
    def forward(ctx, z, weight, node_list):
        zc = _hip.f32_rows(z.detach())
        w = weight.detach()
        nodes = _hip.i64_vec(node_list)
        out = torch.empty((nodes.shape[0], w.shape[1]), dtype=torch.float32, device=zc.device)
        _hip.gemm(zc, w, out, a_rows=nodes)
        ctx.save_for_backward(zc, w, nodes)
        return out

    @staticmethod
    def backward(ctx, g):
        z, w, nodes = ctx.saved_tensors
        g = g.contiguous()
        dz = dw = None
        if ctx.needs_input_grad[0]:
            dz = torch.zeros_like(z).index_add_(0, nodes, g @ w.t())
        if ctx.needs_input_grad[1]:
            dw = z.index_select(0, nodes).t() @ g
        return dz, dw, None
