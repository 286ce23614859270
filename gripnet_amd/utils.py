"""Host-side layout helpers for the supergraph propagation path.

These restate the *input layout contract* the drug-drug (dd) internal layer and
the DistMult decoder rely on (reference: gripnet/utils.py:132-148 `to_bidirection`,
`get_range_list`; :168-198 `process_edge_multirelational`):

  * per relation r the directed edge list is ``cat(fwd_r, reversed fwd_r)``;
  * relations are concatenated in id order;
  * ``edge_type`` repeats r once per directed edge;
  * ``range_list[r] = (start, end)`` is the half-open cumulative range.

They run on the host once per dataset; nothing here touches the GPU.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import weakref

import numpy as np
import torch

from . import _hip

EPS = 1e-13  # reference: gripnet/utils.py:10


def normalize(input: torch.Tensor) -> torch.Tensor:
    """Rows scaled to unit L2 norm (reference: gripnet/utils.py:13-15)."""
    return input / torch.sqrt((input ** 2).sum(dim=1).view(-1, 1))


def sparse_id(n: int) -> torch.Tensor:
    """n x n sparse COO identity, fp32 values, int64 indices, on the CPU (reference: gripnet/utils.py:18-25).

    Every driver builds its node features with it (`GripNet-pose.py:50`, `GripNet-aminer.py:58`); the layers ignore
    that tensor under ``start_graph=True`` (`layers.py:261-262`), so it never reaches a kernel.  Uncoalesced, like the
    reference's ``torch.sparse.FloatTensor(i, v, size)``."""
    idx = torch.arange(int(n), dtype=torch.int64)
    return torch.sparse_coo_tensor(torch.stack([idx, idx]), torch.ones(int(n), dtype=torch.float32), (int(n), int(n)))


def load_graph(pt_file_path: str = "./sample_graph.pt"):
    """``torch.load`` of a pickled graph (reference: gripnet/utils.py:55-80).  The reference's files are pickled
    ``torch_geometric.data.Data`` objects: unpickling them needs torch_geometric installed."""
    return torch.load(pt_file_path, weights_only=False)


def load_node_idx_to_id_dict(pkl_file_path: str = "./data/pose-1/map.pkl"):
    """Index -> entity id map of a dataset directory (reference: gripnet/utils.py:83-95)."""
    import pickle
    with open(pkl_file_path, "rb") as f:
        return pickle.load(f)


def to_bidirection(edge_index: torch.Tensor, edge_type: Optional[torch.Tensor] = None):
    """Append the reversed copy of every edge (reference: gripnet/utils.py:132-138)."""
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    if edge_type is None:
        return both
    return both, torch.cat([edge_type, edge_type])


def remove_bidirection(edge_index: torch.Tensor, edge_type: Optional[torch.Tensor] = None):
    """Keep only edges with src > dst (reference: gripnet/utils.py:122-129)."""
    keep = (edge_index[0] > edge_index[1]).nonzero().view(-1)
    if edge_type is None:
        return edge_index[:, keep]
    return edge_index[:, keep], edge_type[keep]


def get_range_list(parts: Sequence[torch.Tensor], is_node: bool = False) -> torch.Tensor:
    """Cumulative half-open ranges of a list of blocks (reference: gripnet/utils.py:141-148)."""
    axis = 0 if is_node else 1
    sizes = [int(p.shape[axis]) for p in parts]
    ends = np.cumsum(sizes, dtype=np.int64)
    starts = ends - np.asarray(sizes, dtype=np.int64)
    return torch.from_numpy(np.stack([starts, ends], axis=1).reshape(-1, 2).astype(np.int64))


def process_edge(raw_edges: torch.Tensor, rng: Optional[np.random.RandomState] = None):
    """One-relation train / test split (reference: gripnet/utils.py:151-165): keep src > dst, Bernoulli(0.9) per
    undirected edge, both halves back to the bidirectional layout."""
    half = remove_bidirection(raw_edges)
    in_train = (rng or np.random).binomial(1, 0.9, half.shape[1]).astype(bool)
    return (to_bidirection(half[:, torch.from_numpy(np.flatnonzero(in_train))]),
            to_bidirection(half[:, torch.from_numpy(np.flatnonzero(~in_train))]))


def process_node(raw_nodes: torch.Tensor, p: float = 0.9, rng: Optional[np.random.RandomState] = None):
    """Bernoulli train / test split of a node list (reference: gripnet/utils.py:201-209; like the reference the
    draw uses 0.9 whatever ``p`` says)."""
    in_train = (rng or np.random).binomial(1, 0.9, len(raw_nodes)).astype(bool)
    return raw_nodes[torch.from_numpy(np.flatnonzero(in_train))], raw_nodes[torch.from_numpy(np.flatnonzero(~in_train))]


def process_node_multilabel(raw_nodes_list: Sequence[torch.Tensor], rng: Optional[np.random.RandomState] = None):
    """Per-class node split, class-sorted layout and ranges (reference: gripnet/utils.py:212-247).
    Returns ``train_idx, train_class, train_range, test_idx, test_class, test_range``."""
    tr, te = [], []
    for nodes in raw_nodes_list:
        a, b = process_node(nodes, rng=rng)
        tr.append(a)
        te.append(b)

    def classes(parts):
        return torch.cat([torch.full((int(q.shape[0]),), c, dtype=torch.long) for c, q in enumerate(parts)])

    return (torch.cat(tr), classes(tr), get_range_list(tr, is_node=True),
            torch.cat(te), classes(te), get_range_list(te, is_node=True))


def process_edge_multirelational(
    raw_edge_list: Sequence[torch.Tensor],
    p: float = 0.9,
    rng: Optional[np.random.RandomState] = None,
):
    """Bernoulli(p) train/test split per relation, then the bidirectional type-sorted layout.

    Reference: gripnet/utils.py:168-198.  The reference draws from numpy's *global*
    RNG; pass ``rng`` for a private stream (``None`` keeps the global one).
    Returns ``train_idx, train_et, train_range, test_idx, test_et, test_range``.
    """
    draw = (rng or np.random).binomial
    tr_blocks: List[torch.Tensor] = []
    te_blocks: List[torch.Tensor] = []
    tr_types: List[torch.Tensor] = []
    te_types: List[torch.Tensor] = []
    for rel, fwd in enumerate(raw_edge_list):
        in_train = draw(1, p, fwd.shape[1]).astype(bool)
        tr = to_bidirection(fwd[:, torch.from_numpy(np.flatnonzero(in_train))])
        te = to_bidirection(fwd[:, torch.from_numpy(np.flatnonzero(~in_train))])
        tr_blocks.append(tr)
        te_blocks.append(te)
        tr_types.append(torch.full((tr.shape[1],), rel, dtype=torch.long))
        te_types.append(torch.full((te.shape[1],), rel, dtype=torch.long))
    return (
        torch.cat(tr_blocks, dim=1),
        torch.cat(tr_types),
        get_range_list(tr_blocks),
        torch.cat(te_blocks, dim=1),
        torch.cat(te_types),
        get_range_list(te_blocks),
    )


def process_data_multiclass(pairs: torch.Tensor, n_class: int):
    """Group (node, class) pairs by class (reference: gripnet/utils.py:250-263)."""
    nodes, labels, bounds = [], [], [0]
    for c in range(n_class):
        sel = pairs[0][pairs[1] == c]
        nodes.append(sel)
        labels.append(torch.full((sel.shape[0],), c, dtype=torch.int64))
        bounds.append(bounds[-1] + int(sel.shape[0]))
    return torch.cat(nodes), torch.cat(labels), [[bounds[c], bounds[c + 1]] for c in range(n_class)]


def negative_sampling(pos_edge_index: torch.Tensor, num_nodes: int,
                      rng: Optional[np.random.RandomState] = None) -> torch.Tensor:
    """Uniform negative pairs that avoid every positive pair (host side).

    Reference: gripnet/utils.py:98-112.  Same rejection scheme (draw E linear ids in
    [0, n^2), redraw those that hit a positive).  The reference recovers ``row`` with a
    true division followed by ``.long()`` (utils.py:111), which is exact only while
    n^2 < 2^24; integer division is used here instead.
    """
    choice = (rng or np.random).choice
    lin = (pos_edge_index[0] * num_nodes + pos_edge_index[1]).cpu().numpy()
    perm = choice(num_nodes ** 2, lin.shape[0])
    bad = np.flatnonzero(np.isin(perm, lin))
    while bad.size:
        perm[bad] = choice(num_nodes ** 2, bad.size)
        bad = bad[np.isin(perm[bad], lin)]
    perm = torch.from_numpy(perm.astype(np.int64))
    out = torch.stack([perm // num_nodes, perm % num_nodes], dim=0)
    return out.to(pos_edge_index.device)


_samplers = []          # (weak reference to pos_edge_index, _version, range_list contents, num_nodes, NegativeSampler) of the last few positive lists


def typed_negative_sampling(pos_edge_index: torch.Tensor, num_nodes: int, range_list,
                            rng: Optional[np.random.RandomState] = None) -> torch.Tensor:
    """Per-relation negative sampling (reference: gripnet/utils.py:115-119).

    For a positive list that lives on the GPU (what every driver passes, GripNet-pose.py:131) the pairs are drawn ON the
    device: the list gets a `NegativeSampler` the first time it is seen (kept while the same tensor, unmodified, keeps
    coming: the training loop's static `train_idx`), every call draws a fresh seed from numpy's generator (`rng`, or the
    global one the reference seeds at import, utils.py:8-9) - same law as the reference (uniform over the pairs that are
    not positives of the relation), no host round trip, no per-relation Python loop, and the pairs also travel as packed
    32-bit words that the decoder scores at 6 instead of 24 bytes per edge.  CPU tensors take the reference's host loop."""
    if pos_edge_index.is_cuda:
        from ._hip import NegativeSampler
        blocks = tuple((int(s), int(e)) for s, e in torch.as_tensor(range_list).reshape(-1, 2).tolist())   # contents, not identity
        e_total = int(pos_edge_index.shape[1])
        tiles = all(a[1] == b[0] for a, b in zip(blocks, blocks[1:])) and (not blocks or (blocks[0][0] == 0 and blocks[-1][1] == e_total)) \
            and all(s <= e for s, e in blocks)
        if not tiles:
            # blocks that overlap, leave gaps or come out of order: the reference samples each [start, end) on its own and
            # concatenates (utils.py:115-119) - one sampler per block, nothing cached
            parts = [NegativeSampler(pos_edge_index[:, s:e].contiguous(), num_nodes).sample(
                seed=int((rng or np.random).randint(0, 2 ** 31 - 1))) for s, e in blocks]
            return torch.cat(parts, dim=1) if parts else pos_edge_index[:, :0].clone()
        hit = None
        for entry in _samplers:
            if entry[0]() is pos_edge_index and entry[1] == pos_edge_index._version and entry[2] == blocks and entry[3] == num_nodes:
                hit = entry
                break
        _samplers[:] = [en for en in _samplers if en[0]() is not None]     # lists that are gone take their samplers (and bitmaps) with them
        if hit is None:
            hit = (weakref.ref(pos_edge_index), pos_edge_index._version, blocks, num_nodes,
                   NegativeSampler(pos_edge_index, num_nodes, range_list))
            _samplers.insert(0, hit)
            del _samplers[3:]
        seed = int((rng or np.random).randint(0, 2 ** 31 - 1))
        return hit[4].sample(seed=seed)
    parts = [negative_sampling(pos_edge_index[:, int(s):int(e)], num_nodes, rng) for s, e in range_list]
    return torch.cat(parts, dim=1)


def device_negative_sampler(pos_edge_index: torch.Tensor, num_nodes: int, range_list=None):
    """GPU counterpart of `typed_negative_sampling` for a static positive edge list: build once, then
    ``sampler.sample(seed)`` every epoch (no host round trip, no per-relation Python loop)."""
    from ._hip import NegativeSampler
    return NegativeSampler(pos_edge_index, num_nodes, range_list)


def relation_metrics(pos_score: torch.Tensor, neg_score: torch.Tensor, range_list):
    """Per-relation (auprc, auroc, ap) on the GPU, float64 [R] each: what the reference's epoch loop computes
    with one `auprc_auroc_ap` call per relation (GripNet-pose.py:148-160).

    The metrics are where an epoch synchronises anyway, so this is also where an out-of-range id seen by the decoder
    kernels since the last check becomes the reference's IndexError (the kernels write NaN scores and set a flag; see
    multiRelaInnerProductDecoder)."""
    from ._hip import link_metrics, raise_if_index_errors
    raise_if_index_errors(pos_score.device)
    return link_metrics(pos_score, neg_score, range_list)


def link_loss(pos_score: torch.Tensor, neg_score: torch.Tensor, eps: float = EPS) -> torch.Tensor:
    """``-log(pos + EPS).mean() - log(1 - neg + EPS).mean()``: the training loss of GripNet-pose.py:140-142 as one launch
    forward and one backward (gn_link_loss_*), differentiable.  The reference spells the expression out with torch ops
    in its driver; that keeps working - this is the same value for callers that want it in two launches instead of ~20."""
    from .autograd import LinkLossFn
    _hip.require_gpu(pos_score, neg_score)
    return LinkLossFn.apply(pos_score, neg_score, float(eps))


def link_prediction_loss(decoder, z: torch.Tensor, pos_index: torch.Tensor, neg_index: torch.Tensor, edge_type: torch.Tensor,
                         eps: float = EPS):
    """``loss, pos_score, neg_score``: the two decoder calls and the loss of a training step (GripNet-pose.py:137-142:
    ``pos_score = dmt(z, pos_index, et); neg_score = dmt(z, neg_index, et); loss = -log(pos + EPS).mean() - log(1 - neg + EPS).mean()``)
    as one autograd node.  Values and gradients are those of the three separate calls (bit for bit where the fused launches
    apply); the backward pass makes two decoder launches that compute the loss's derivative inline instead of a loss launch, two
    [E] gradient vectors and two decoder launches.  The scores are returned for the metrics and carry no gradient.  The
    reference's spelled-out expression keeps working; this is the build's own counterpart of those three lines."""
    from .autograd import LinkPredictionLossFn
    _hip.require_gpu(z, pos_index, neg_index, edge_type, decoder.weight)
    if z.shape[1] != decoder.in_dim:
        raise ValueError("expected {} features, got {}".format(decoder.in_dim, z.shape[1]))
    plan = decoder.plan_for(z, pos_index, edge_type)
    if plan is not None and plan.num_nodes != z.shape[0]:
        plan = None
    return LinkPredictionLossFn.apply(z, decoder.weight, pos_index, neg_index, edge_type, float(eps), plan)


def class_loss(score: torch.Tensor, classes: torch.Tensor, eps: float = EPS) -> torch.Tensor:
    """``-log(score[range(n), classes] + EPS).mean()``: the training loss of GripNet-aminer.py:133 (and of every freebase
    driver) as one launch forward and one backward (gn_class_loss_*), differentiable.  The drivers spell it out with torch
    advanced indexing; that keeps working - this is the same value without the index / log / mean kernels."""
    from .autograd import ClassLossFn
    _hip.require_gpu(score, classes)
    return ClassLossFn.apply(score, classes, float(eps))


def forget_call_memos(module) -> int:
    """Drop the recorded call sequences (``_hip.CallMemo``) of every layer / decoder under `module`: the steady-state shortcut
    of the inference forwards keeps the plans and graph tensors its recordings name alive (at most eight entries per module) -
    call this after switching to another dataset to release them at once.  Returns the number of memos dropped."""
    dropped = 0
    for m in module.modules():
        if m.__dict__.pop("_memo", None) is not None:
            dropped += 1
    return dropped


def set_table_storage(module, storage: str = "bf16"):
    """Switch every GCN-style layer under `module` to "bf16" (or back to "fp32") storage of its gathered table:
    x W is rounded to bf16 once per forward and read at half the bytes; sums, bias, activation, outputs and every
    parameter stay fp32.  Under training (round 6) the forward takes the rounded table too and the backward is the fp32
    layer's (the rounding's straight-through derivative).  The reference has no reduced precision; this is
    the build's own variant for the node-classification suite.  Returns the layers touched."""
    from .layers import myGCN
    if storage not in ("fp32", "bf16"):
        raise ValueError("storage must be 'fp32' or 'bf16', got {!r}".format(storage))
    touched = [m for m in module.modules() if isinstance(m, myGCN)]
    for m in touched:
        m.table_storage = storage
    return touched


def set_arithmetic(module, arithmetic: str = "fp32"):
    """Arithmetic of the dense products of every layer under `module`: "fp32" (default) is fp32-faithful - three-term
    bf16 splits (what is dropped is below 2^-23 of a product) or the fp32 matrix instruction, the reference's
    precision; "fast" keeps two bf16 terms (<= 2^-16 per product).  The mode travels with every call as a flag of the
    C ABI.  Returns the layers touched."""
    from .layers import myGCN, myRGCN
    if arithmetic not in ("fp32", "fast"):
        raise ValueError("arithmetic must be 'fp32' or 'fast', got {!r}".format(arithmetic))
    touched = [m for m in module.modules() if isinstance(m, (myGCN, myRGCN))]
    for m in touched:
        m.arithmetic = arithmetic
    return touched


def profile(fn):
    """No-op stand-in for ``pytorch_memlab.profile`` (reference: GripNet-pose.py:18,112)."""
    return fn


def shard_edge_ranges(num_edges: int, world_size: int) -> List[Tuple[int, int]]:
    """Split ``[0, num_edges)`` into ``world_size`` contiguous ranges balanced by edge count.

    Boundaries may fall inside a relation; the type-sorted layout keeps each shard a
    contiguous slice of ``edge_index``.
    """
    base, extra = divmod(int(num_edges), int(world_size))
    out, lo = [], 0
    for k in range(world_size):
        hi = lo + base + (1 if k < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


# ---- evaluation metrics (host side, scikit-learn; reference: gripnet/utils.py:28-52) ----------------
def _to_numpy(*tensors):
    from ._hip import raise_if_index_errors
    for t in tensors:                                   # the copy synchronises: a natural place for the deferred id check
        if t.is_cuda:
            raise_if_index_errors(t.device)
    return [t.detach().cpu().numpy() for t in tensors]


def auprc_auroc_ap(target_tensor: torch.Tensor, score_tensor: torch.Tensor):
    """(area under the precision-recall curve, ROC AUC, average precision) of binary link scores."""
    from sklearn import metrics
    y, pred = _to_numpy(target_tensor, score_tensor)
    precision, recall, _ = metrics.precision_recall_curve(y, pred)
    return metrics.auc(recall, precision), metrics.roc_auc_score(y, pred), metrics.average_precision_score(y, pred)


def micro_macro(target_tensor: torch.Tensor, score_tensor: torch.Tensor):
    """(micro-F1, macro-F1) of predicted class ids."""
    from sklearn import metrics
    y, pred = _to_numpy(target_tensor, score_tensor)
    return metrics.f1_score(y, pred, average="micro"), metrics.f1_score(y, pred, average="macro")


def acc(target_tensor: torch.Tensor, score_tensor: torch.Tensor):
    from sklearn.metrics import accuracy_score
    y, pred = _to_numpy(target_tensor, score_tensor)
    return accuracy_score(y, pred)
