"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.

What runs here is the product's orchestration (gripnet_amd/sharded.py: edge-range sharding, the
one all-reduce, finalisation, per-rank decoder slices).  The arithmetic behind it is injected: the
CPU oracle stands in for the HIP kernels (there is no GPU in this container and the product has no
CPU fallback), which is the oracle's job as a checker.  The result of the 2-rank run must equal the
oracle's single-process forward.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardKernels:
    """Same four operations as gripnet_amd.sharded.HipShardKernels, computed by the CPU oracle."""

    def __init__(self, sd, data, lo, hi):
        from oracle import gripnet_oracle as orc
        self.orc, self.sd, self.data, self.lo, self.hi = orc, sd, data, lo, hi

    def encode_genes(self):
        o, d = self.orc, self.data
        z = o.homo_forward(self.sd, "gg.", None, d.gg_edge_index, d.edge_weight, if_catout=True)
        return o.inter_forward(self.sd, "gd.", z, d.gd_edge_index, None, if_relu=True, mod="cat")

    def _partial_sum(self, x, basis, att):
        # un-normalised sum over my edge range: relation of an edge = range_list row that holds it (differentiable)
        d = self.data
        w = (att @ basis.reshape(basis.shape[0], -1)).view(att.shape[0], basis.shape[1], basis.shape[2])
        out = torch.zeros((x.shape[0], basis.shape[2]))
        for r in range(d.train_range.shape[0]):
            s, e = max(int(d.train_range[r, 0]), self.lo), min(int(d.train_range[r, 1]), self.hi)
            if s < e:
                out = out.index_add(0, d.train_idx[1, s:e], x.index_select(0, d.train_idx[0, s:e]) @ w[r])
        return out

    def partial(self, x, out, fresh_weights=False):
        sd = self.sd
        with torch.no_grad():
            out.copy_(self._partial_sum(x, sd["dd.conv_list.0.basis"], sd["dd.conv_list.0.att"]))
        return out

    # ---- training: the same operations with gradients (torch autograd through the oracle's ops) ----
    def edge_gradients(self, x, gm):
        sd = self.sd
        xs, bs, at = (t.detach().clone().requires_grad_() for t in (x, sd["dd.conv_list.0.basis"], sd["dd.conv_list.0.att"]))
        with torch.enable_grad():
            p = self._partial_sum(xs, bs, at)
        return torch.autograd.grad(p, (xs, bs, at), gm)

    def rgcn_parameters(self):
        sd = self.sd
        return sd["dd.conv_list.0.basis"], sd["dd.conv_list.0.att"], sd["dd.conv_list.0.root"], None

    def in_degree(self):
        d = self.data
        n = int(d.n_d_node)
        return torch.zeros(n).index_add_(0, d.train_idx[1], torch.ones(d.train_idx.shape[1])).clamp(min=1)

    def score_edges(self, z, edge_index, sigmoid=True):
        return self.orc.distmult(z, edge_index, self.data.train_et[self.lo:self.hi], self.sd["dmt.weight"], sigmoid=sigmoid)

    # the element-wise glue of the sharded training step (the product's kernels do these on the library's launches)
    def layer_gradient(self, g, out, want_bias):
        g = (g * (out > 0).to(g.dtype)).contiguous()                   # ReLU mask by the saved output
        return g, g / self.in_degree().view(-1, 1), (g.sum(dim=0) if want_bias else None)

    def root_gradients(self, g, root, x, dx):
        return dx + g @ root.t(), x.t() @ g

    def shard_loss(self, pos, neg, total_edges, eps):
        return -(torch.log(pos + eps).sum() + torch.log(1 - neg + eps).sum()) / float(total_edges)

    def parameters(self):
        return list(self.sd.values())

    def decoder_weight(self):
        return self.sd["dmt.weight"]

    def finalize(self, summed, x, out, slot0):
        d = self.data
        n = x.shape[0]
        cnt = torch.zeros(n).index_add_(0, d.train_idx[1], torch.ones(d.train_idx.shape[1]))
        with torch.no_grad():
            out.copy_(torch.relu(summed / cnt.clamp(min=1).view(-1, 1) + x @ self.sd["dd.conv_list.0.root"]))
            if slot0 is not None:
                slot0.copy_(x)
        return out

    def score(self, z, sigmoid=True):
        d = self.data
        return self.orc.distmult(z, d.train_idx[:, self.lo:self.hi], d.train_et[self.lo:self.hi],
                                 self.sd["dmt.weight"], sigmoid=sigmoid)

    # the decoder in two parts (the input columns of z beside the exchange, the rest after it)
    def _partial_scores(self, t, c0, c1):
        d = self.data
        idx, et = d.train_idx[:, self.lo:self.hi], d.train_et[self.lo:self.hi]
        w = self.sd["dmt.weight"].detach()[:, c0:c1]
        return (t[idx[0], c0:c1] * t[idx[1], c0:c1] * w[et]).sum(1)

    def score_input_columns(self, x, sigmoid=True):
        self.calls = getattr(self, "calls", 0) + 1
        return self._partial_scores(x, 0, x.shape[1]), x.shape[1]

    def score_rest(self, started, z, sigmoid=True):
        part, col = started
        s = part + self._partial_scores(z, col, z.shape[1])
        return torch.sigmoid(s) if sigmoid else s


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gripnet_amd.pipeline import PoseModel
        from gripnet_amd.sharded import ShardedPoseForward
        from gripnet_amd.synth import make_pose
        from gripnet_amd.utils import shard_edge_ranges
        data = make_pose("small")
        torch.manual_seed(1111)
        model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        lo, hi = shard_edge_ranges(int(data.train_idx.shape[1]), world)[rank]
        fwd = ShardedPoseForward(model, data, rank, world, kernels=OracleShardKernels(sd, data, lo, hi))
        assert (fwd.edge_lo, fwd.edge_hi) == (lo, hi)
        z, score = fwd()
        assert fwd.kernels.calls == 1                         # world_size > 1: the input columns were scored beside the exchange
        # every rank holds the same z; rank 0 gathers the score slices in rank order
        zs = [torch.empty_like(z) for _ in range(world)]
        dist.all_gather(zs, z)
        sizes = [b - a for a, b in shard_edge_ranges(int(data.train_idx.shape[1]), world)]
        parts = [torch.empty(s) for s in sizes]
        if rank == 0:
            parts[0] = score
            for r in range(1, world):
                dist.recv(parts[r], src=r)
            # numpy arrays travel by value: a tensor would travel as a shared-memory handle that dies
            # with this process if the parent has not mapped it yet
            q.put((z.numpy(), [t.numpy() for t in zs], torch.cat(parts).numpy(),
                   {k: v.numpy() for k, v in sd.items()}, lo, hi))
        else:
            dist.send(score.contiguous(), dst=0)
    finally:
        dist.destroy_process_group()


def _train_worker(rank, world, port, q, steps=1):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gripnet_amd.pipeline import PoseModel
        from gripnet_amd.sharded import ShardedPoseTraining
        from gripnet_amd.synth import make_pose
        from gripnet_amd.utils import shard_edge_ranges
        data = make_pose("small")
        torch.manual_seed(1111)
        model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
        sd = {k: v.detach().clone().requires_grad_() for k, v in model.state_dict().items()}
        lo, hi = shard_edge_ranges(int(data.train_idx.shape[1]), world)[rank]
        neg = torch.randint(0, data.n_d_node, data.train_idx.shape, generator=torch.Generator().manual_seed(5))
        step = ShardedPoseTraining(model, data, rank, world, kernels=OracleShardKernels(sd, data, lo, hi))
        loss = step.step(neg)
        for _ in range(steps - 1):                                 # gradient accumulation: same inputs, grads add up
            loss = step.step(neg, zero_grad=False)
        q.put((rank, float(loss), {k: (None if v.grad is None else v.grad.numpy()) for k, v in sd.items()}))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_rank_sharded_forward_equals_single_process():
    sys.path.insert(0, REPO)
    from gripnet_amd.synth import make_pose
    from oracle import gripnet_oracle as orc
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    z, zs, score, sd, lo, hi = q.get(timeout=240)
    z, zs, score = torch.from_numpy(z), [torch.from_numpy(t) for t in zs], torch.from_numpy(score)
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = make_pose("small")
    ref = orc.pose_forward(sd, data.gg_edge_index, data.edge_weight, data.gd_edge_index, data.train_idx,
                           data.train_et, data.train_range)
    assert (lo, hi) == (0, (data.train_idx.shape[1] + 1) // 2)
    for other in zs:
        assert torch.equal(other, z)                          # replicated result, bit for bit after the all-reduce
    assert (z - ref["z_dd"]).abs().max().item() <= 1e-5       # sum order differs from 1 rank: not bit-exact
    assert score.shape == ref["score"].shape
    assert (score - ref["score"]).abs().max().item() <= 1e-5


def test_world_size_one_needs_no_process_group():
    sys.path.insert(0, REPO)
    from gripnet_amd.pipeline import PoseModel
    from gripnet_amd.sharded import ShardedPoseForward
    from gripnet_amd.synth import make_pose
    from oracle import gripnet_oracle as orc
    data = make_pose("tiny")
    torch.manual_seed(3)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    E = int(data.train_idx.shape[1])
    fwd = ShardedPoseForward(model, data, 0, 1, kernels=OracleShardKernels(sd, data, 0, E))
    z, score = fwd(sigmoid=False)
    ref = orc.pose_forward(sd, data.gg_edge_index, data.edge_weight, data.gd_edge_index, data.train_idx,
                           data.train_et, data.train_range, sigmoid=False)
    assert (z - ref["z_dd"]).abs().max().item() <= 1e-5
    assert (score - ref["score"]).abs().max().item() <= 1e-5


@pytest.mark.timeout(300)
@pytest.mark.parametrize("steps", [1, 2])
def test_two_rank_sharded_training_step_equals_single_process(steps):
    """Forward + backward with the dd edges on two ranks: the loss and every parameter gradient equal the
    single-process ones (torch autograd through the oracle), on BOTH ranks.  steps = 2: a second step with
    zero_grad=False accumulates - every gradient doubles, the decoder weight's included (its exchange must carry
    only the new step's share)."""
    sys.path.insert(0, REPO)
    from gripnet_amd.pipeline import PoseModel
    from gripnet_amd.synth import make_pose
    from oracle import gripnet_oracle as orc
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, steps)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = make_pose("small")
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
    sd = {k: v.detach().clone().requires_grad_() for k, v in model.state_dict().items()}
    neg = torch.randint(0, data.n_d_node, data.train_idx.shape, generator=torch.Generator().manual_seed(5))
    ref = orc.pose_forward(sd, data.gg_edge_index, data.edge_weight, data.gd_edge_index, data.train_idx,
                           data.train_et, data.train_range)
    neg_score = orc.distmult(ref["z_dd"], neg, data.train_et, sd["dmt.weight"])
    loss = -torch.log(ref["score"] + 1e-13).mean() - torch.log(1 - neg_score + 1e-13).mean()
    loss.backward()
    unused = {"gd.target_feat_down"}                              # exists in the model, unused in cat mode
    for rank, rank_loss, grads in got:
        assert abs(rank_loss - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
        for k, v in sd.items():
            if k in unused:
                continue
            assert grads[k] is not None, k
            err = (torch.from_numpy(grads[k]) - steps * v.grad).abs().max().item()
            scale = steps * max(1.0, v.grad.abs().max().item())
            assert err <= 2e-5 * scale, "rank {} {}: {:.3e}".format(rank, k, err)
