#!/usr/bin/env python3
"""Plan-level simulation of the relational kernel's padding (development tool, CPU only).

    python tools/acc_sim.py [pose0-syn]

For every (relation, pool of destination rows) it packs the rows' edge lists into gather tiles of 16 rows - a row may
be split over several gather rows ("pieces") - sorted by piece length, and prices the result with the measured costs
of k_rgcn_acc (per non-empty tile, per gather iteration).  pool = 16: pieces stay inside their 16-row tile; pool = 48:
the three tiles a wave owns are packed together (needs the 0/1 matrix step `acc += P (S W_r)` to put the sums back
on their accumulator rows).  gran = iterations per stream block.  The row "current" is what the shipped plan does
(one gather row per destination row, blocks of four iterations).  See DESIGN.md section 4.1.
"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from gripnet_amd.synth import make_pose
name = sys.argv[1] if len(sys.argv) > 1 else 'pose0-syn'
d = make_pose(name)
n, R = d.n_d_node, d.n_dd_edge_type
dst = d.train_idx[1].numpy(); rel = d.train_et.numpy()
cnt = np.zeros((R, n), dtype=np.int32)
np.add.at(cnt, (rel, dst), 1)
E = len(dst)
CAP = 32
def best(c, G, TILE, ITER):
    c = c[c > 0]
    if len(c) == 0: return 0, 0, 0.
    bestc = None
    Lmax = min(int(c.max()), CAP)
    for L in range(G, Lmax + G, G):
        full = c // L; rem = c % L
        pieces = np.concatenate([np.repeat(L, int(full.sum())), rem[rem > 0]])
        pieces = -np.sort(-pieces)
        T = (len(pieces) + 15) // 16
        its = (np.ceil(pieces[::16] / G) * G).sum()
        cost = T * TILE + its * ITER
        if bestc is None or cost < bestc[2]: bestc = (T, int(its), cost)
        if L > 4 * G and cost > 1.5 * bestc[2]: break
    return bestc
for P in (16, 48):
  for G in (1, 2, 4):
    pools = [(a, min(n, a + P)) for a in range(0, n, P)]
    T = I = 0; C = 0.
    for r in range(R):
        for a, b in pools:
            t, i, c = best(cnt[r, a:b], G, 800., 231.)
            T += t; I += i; C += c
    print('pool', P, 'gran', G, 'tiles', T, 'iters', I, 'iters/ideal', round(I / (E / 16), 3), 'cost(M)', round(C / 1e6, 1), ' [current: 39521 tiles x 400 + 294280 x 231 =', round((39521*400 + 294280*231)/1e6, 1), ']')
