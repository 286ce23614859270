"""Synthetic supergraphs with the field names the reference drivers consume.

The real PoSE / aminer / freebase files are Dropbox downloads that are absent here
(reference README.md:37-51), so every workload is synthetic.  The generator and the
configuration ladder follow SURVEY.md section 8(d); field names follow the pickled
``Data`` objects the drivers read (reference: GripNet-pose.py:50-55,95-98,117-131;
GripNet-aminer.py:47-65,104-107; GripNet-freebase-c.py:59-63,150-157).
"""
from __future__ import annotations

from typing import Dict

import torch

from .utils import get_range_list, to_bidirection


class Data:
    """Minimal attribute bag standing in for ``torch_geometric.data.Data``.

    Only what the drivers use: attribute access, ``from_dict`` and ``.to(device)``
    (reference: GripNet-pose.py:60,66-67).
    """

    def __init__(self, **fields):
        for k, v in fields.items():
            setattr(self, k, v)

    @classmethod
    def from_dict(cls, d: Dict):
        return cls(**d)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def to(self, device):
        for k in self.keys():
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self

    def __repr__(self):
        parts = []
        for k in self.keys():
            v = getattr(self, k)
            parts.append("{}={}".format(k, list(v.shape) if torch.is_tensor(v) else v))
        return "Data({})".format(", ".join(parts))


# name -> (n_g, E_gg_dir, n_d, E_gd, R, E_dd_dir)   (SURVEY.md section 8d ladder)
POSE_LADDER = {
    "tiny": dict(n_g=50, e_gg_dir=200, n_d=12, e_gd=40, n_rel=5, e_dd_dir=None),
    "small": dict(n_g=2000, e_gg_dir=20000, n_d=128, e_gd=2000, n_rel=32, e_dd_dir=25000),
    "pose0-syn": dict(n_g=19081, e_gg_dir=715612, n_d=645, e_gd=18690, n_rel=964, e_dd_dir=1_000_000),
    "pose1-syn": dict(n_g=19081, e_gg_dir=715612, n_d=645, e_gd=18690, n_rel=964, e_dd_dir=2_000_000),
    "pose2-syn": dict(n_g=19081, e_gg_dir=715612, n_d=645, e_gd=18690, n_rel=964, e_dd_dir=4_200_000),
}


def _relation_sizes(n_rel: int, e_dd_dir):
    if e_dd_dir is None:  # the "tiny" rule: s_r = 7 + r, r = 1..R
        return [7 + r for r in range(1, n_rel + 1)]
    ranks = torch.arange(1, n_rel + 1, dtype=torch.float64)
    w = ranks.pow(-0.8)
    sizes = torch.floor(e_dd_dir * w / w.sum()).clamp_(min=1).to(torch.int64)
    return sizes.tolist()


def make_pose(name: str = "tiny", seed: int = 7, dd_scale: int = 1, **override) -> Data:
    """Synthetic PoSE-like supergraph: gene graph gg, gene->drug gd, multi-relational dd.

    ``dd_scale`` multiplies the directed dd edge budget (used for weak-scaling runs where
    the global dd edge list grows with the number of GPUs).
    """
    cfg = dict(POSE_LADDER[name])
    cfg.update(override)
    if cfg["e_dd_dir"] is not None:
        cfg["e_dd_dir"] = int(cfg["e_dd_dir"]) * int(dd_scale)
    g = torch.Generator().manual_seed(seed)
    n_g, n_d, n_rel = cfg["n_g"], cfg["n_d"], cfg["n_rel"]

    a = torch.randint(0, n_g, (2, cfg["e_gg_dir"]), generator=g)
    a = a[:, a[0] != a[1]]
    gg = to_bidirection(a)  # duplicates kept on purpose

    gd = torch.stack([
        torch.randint(0, n_g, (cfg["e_gd"],), generator=g),
        torch.randint(0, n_d, (cfg["e_gd"],), generator=g),
    ])

    blocks = []
    for s_r in _relation_sizes(n_rel, cfg["e_dd_dir"]):
        e = torch.randint(0, n_d, (2, int(s_r)), generator=g)
        blocks.append(to_bidirection(e[:, e[0] != e[1]]))
    train_idx = torch.cat(blocks, dim=1)
    train_range = get_range_list(blocks)
    train_et = torch.repeat_interleave(
        torch.arange(n_rel, dtype=torch.long), train_range[:, 1] - train_range[:, 0])

    return Data(
        name=name,
        n_g_node=n_g, n_d_node=n_d, n_gg_edge=int(gg.shape[1]), n_dd_edge_type=n_rel,
        gg_edge_index=gg.long(), gd_edge_index=gd.long(),
        edge_weight=torch.ones(gg.shape[1]),
        train_idx=train_idx.long(), train_et=train_et, train_range=train_range,
    )


def add_pose_test_split(data: Data, seed: int = 13, ratio: int = 9) -> Data:
    """Adds ``test_idx / test_et / test_range`` to a synthetic PoSE graph: per relation one held-out pair for every `ratio`
    training pairs (the reference splits every relation's pairs 90 / 10, gripnet/utils.py:168-198 with p = 0.9), in the same
    layout as the training list (both directions, type-sorted, cumulative ranges).  What test() of GripNet-pose.py:180-201 scores."""
    g = torch.Generator().manual_seed(seed)
    n_d = int(data.n_d_node)
    sizes = ((data.train_range[:, 1] - data.train_range[:, 0]) // 2).tolist()
    blocks = []
    for s_r in sizes:
        k = max(1, int(s_r) // ratio)
        e = torch.randint(0, n_d, (2, k), generator=g)
        e = e[:, e[0] != e[1]]
        if e.shape[1] == 0:                           # (a relation keeps at least one test pair)
            e = torch.tensor([[0], [1 % n_d]])
        blocks.append(to_bidirection(e))
    data.test_idx = torch.cat(blocks, dim=1).long()
    data.test_range = get_range_list(blocks)
    data.test_et = torch.repeat_interleave(torch.arange(len(blocks), dtype=torch.long),
                                           data.test_range[:, 1] - data.test_range[:, 0])
    return data


def pose_edges_aggregated(data: Data) -> int:
    """Numerator of the headline metric: A = 2(E_gg + n_g) + E_gd + E_dd."""
    return (2 * (int(data.gg_edge_index.shape[1]) + int(data.n_g_node))
            + int(data.gd_edge_index.shape[1]) + int(data.train_idx.shape[1]))


def make_rgcn_pose(name: str = "pose0-syn", seed: int = 7, n_rel: int = None, **override) -> Data:
    """The homogenised graph of the reference's all-nodes baselines (baselines/LP_baselines/rgcn_pose.py:28-35,91-106,
    `pose-*-combl.pt`): ONE node set of n_d drugs followed by n_g genes, relations = the drug-drug relations, then the
    gene-gene edges, then the gene-drug edges as the LAST TWO relations (the caller slices `train_range[:-2]` for the
    drug-only negatives), every relation bidirectional and type-sorted.  Built from the PoSE ladder's graph; `n_rel`
    keeps the total at that many relations (default: the ladder's count, so that R matches the GripNet runs)."""
    base = make_pose(name, seed=seed, **override)
    n_d, n_g = int(base.n_d_node), int(base.n_g_node)
    R = int(base.n_dd_edge_type) if n_rel is None else int(n_rel)
    rng = base.train_range
    keep = R - 2                                          # drug-drug relations kept (the largest ones: the ladder is Zipf-ordered)
    blocks = [base.train_idx[:, int(rng[r, 0]):int(rng[r, 1])] for r in range(keep)]
    blocks.append(base.gg_edge_index + n_d)               # gene ids follow the drug ids
    gd = torch.stack([base.gd_edge_index[0] + n_d, base.gd_edge_index[1]])
    blocks.append(to_bidirection(gd))
    train_idx = torch.cat(blocks, dim=1)
    train_range = get_range_list(blocks)
    train_et = torch.repeat_interleave(torch.arange(R, dtype=torch.long), train_range[:, 1] - train_range[:, 0])
    return Data(name="rgcn-" + name, n_node=n_d + n_g, n_drug=n_d, n_edge_type=R,
                train_idx=train_idx.long(), train_et=train_et, train_range=train_range)


NC_LADDER = {
    "tiny": dict(n_p=60, e_pp=150, n_q=40, e_qq=90, n_a=25, e_pa=80, e_qa=60, e_aa=50, n_class=4),
    "aminer-syn": dict(n_p=50_000, e_pp=250_000, n_q=30_000, e_qq=150_000, n_a=20_000,
                       e_pa=150_000, e_qa=100_000, e_aa=100_000, n_class=8),
}


def make_nc(name: str = "tiny", seed: int = 11, **override) -> Data:
    """Synthetic node-classification supergraph (aminer / freebase-b/c/d field names)."""
    cfg = dict(NC_LADDER[name])
    cfg.update(override)
    g = torch.Generator().manual_seed(seed)

    def homo(n, e_dir):
        a = torch.randint(0, n, (2, e_dir), generator=g)
        return to_bidirection(a[:, a[0] != a[1]]).long()

    def bip(n_src, n_dst, e):
        return torch.stack([torch.randint(0, n_src, (e,), generator=g),
                            torch.randint(0, n_dst, (e,), generator=g)]).long()

    pp, qq, aa = homo(cfg["n_p"], cfg["e_pp"]), homo(cfg["n_q"], cfg["e_qq"]), homo(cfg["n_a"], cfg["e_aa"])
    labels = torch.randint(0, cfg["n_class"], (cfg["n_a"],), generator=g)
    return Data(
        name=name,
        n_p_node=cfg["n_p"], n_q_node=cfg["n_q"], n_a_node=cfg["n_a"], n_a_type=cfg["n_class"],
        n_pp_edge=int(pp.shape[1]), n_qq_edge=int(qq.shape[1]), n_aa_edge=int(aa.shape[1]),
        pp_edge_idx=pp, qq_edge_idx=qq, aa_edge_idx=aa,
        pa_edge_idx=bip(cfg["n_p"], cfg["n_a"], cfg["e_pa"]),
        qa_edge_idx=bip(cfg["n_q"], cfg["n_a"], cfg["e_qa"]),
        pp_edge_weight=torch.ones(pp.shape[1]), qq_edge_weight=torch.ones(qq.shape[1]),
        aa_edge_weight=torch.ones(aa.shape[1]),
        a_label=labels,
    )
