#!/usr/bin/env python3
"""Counterpart of the reference's link-prediction driver (GripNet-pose.py) on the MI355X path.

The reference script executes at import, loads a pickled torch_geometric Data object from a Dropbox
download (GripNet-pose.py:40) and trains full-batch for 100 epochs.  The datasets are not available
offline, so this driver runs the same model, loss, optimiser and epoch structure
(GripNet-pose.py:86-104,112-172,180-201) on the synthetic PoSE ladder of gripnet_amd.synth:

    python examples/train_pose.py --workload small --epochs 20

Every layer call below is the reference's; only the import line differs.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gripnet_amd.decoder import multiRelaInnerProductDecoder         # reference: from gripnet.decoder import ...
from gripnet_amd.layers import homoGraph, interGraph                  # reference: from gripnet.layers import ...
from gripnet_amd.synth import make_pose
from gripnet_amd.utils import EPS, device_negative_sampler, relation_metrics


class Model(torch.nn.Module):                                         # GripNet-pose.py:73-82
    def __init__(self, gg, gd, dd, dmt):
        super().__init__()
        self.gg, self.gd, self.dd, self.dmt = gg, gd, dd, dmt

    def forward(self, *args):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="small")
    ap.add_argument("--epochs", type=int, default=20)
    args = ap.parse_args()
    torch.manual_seed(1111)
    np.random.seed(1111)
    device = torch.device("cuda")
    data = make_pose(args.workload).to(device)
    n_g, n_d, n_et = data.n_g_node, data.n_d_node, data.n_dd_edge_type
    gg_nhids, gd_out, dd_nhids = [32, 16, 16], [16, 32], [48, 32]      # GripNet-pose.py:86-89
    model = Model(homoGraph(gg_nhids, start_graph=True, in_dim=n_g),
                  interGraph(sum(gg_nhids), gd_out[0], n_d, target_feat_dim=gd_out[-1]),
                  homoGraph(dd_nhids, multi_relational=True, n_rela=n_et),
                  multiRelaInnerProductDecoder(sum(dd_nhids), n_et)).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01)         # GripNet-pose.py:104
    # the reference calls typed_negative_sampling(train_idx, n_d, train_range) every epoch (host numpy,
    # one Python iteration per relation, GripNet-pose.py:131); same distribution, drawn on the GPU
    sampler = device_negative_sampler(data.train_idx, n_d, data.train_range)

    def train(epoch):                                                 # GripNet-pose.py:112-172
        model.train()
        optimizer.zero_grad()
        z = model.gg(None, data.gg_edge_index, edge_weight=data.edge_weight, if_catout=True)
        z = model.gd(z, data.gd_edge_index, mod="cat", if_relu=True)
        z = model.dd(z, data.train_idx, edge_type=data.train_et, range_list=data.train_range, if_catout=True)
        pos_index = data.train_idx
        neg_index = sampler.sample(seed=epoch)
        pos_score = model.dmt(z, pos_index, data.train_et)
        neg_score = model.dmt(z, neg_index, data.train_et)
        loss = -torch.log(pos_score + EPS).mean() - torch.log(1 - neg_score + EPS).mean()
        loss.backward()
        optimizer.step()
        # per-relation AUPRC / AUROC / AP, averaged over relations: the reference loops over the relations
        # with one scikit-learn call each (GripNet-pose.py:148-160); here one pass on the GPU
        auprc, auroc, ap = relation_metrics(pos_score, neg_score, data.train_range)
        return z.detach(), float(loss), float(auprc.nanmean()), float(auroc.nanmean()), float(ap.nanmean())

    for epoch in range(args.epochs):
        t0 = time.time()
        z, loss, auprc, auroc, ap = train(epoch)
        torch.cuda.synchronize()
        print("{:3d} loss:{:0.4f}   auprc:{:0.4f}   auroc:{:0.4f}   ap@50:{:0.4f}   time:{:0.2f}s".format(
            epoch, loss, auprc, auroc, ap, time.time() - t0))


if __name__ == "__main__":
    main()
