"""Shared test helpers (CPU side).  The oracle is imported here because tests are its allowed users."""
from __future__ import annotations

import os

import numpy as np
import torch

from dgnn_amd.config import Config, reconbench_pretrained

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def kf96_state_dict():
    w = gold("kf96_weights.npz")
    return {k: torch.from_numpy(w[k]) for k in w.files}


def oracle_static(train=False, dtype=torch.float32, convs=(64, 128, 128, 128), load=True, seed=0):
    from oracle.static_edge_filters import SurfaceNet
    clf = reconbench_pretrained(device="cpu", convs=convs)
    torch.manual_seed(seed)
    net = SurfaceNet(clf)
    if load:
        net.load_state_dict(kf96_state_dict())
    net = net.to(dtype)
    return net.train() if train else net.eval()


def f3_data(g):
    """Rebuild the reference `data` object of SurfaceNet.forward from the F3 fixture."""
    adjs = []
    i = 0
    while "adj%d_edge_index" % i in g.files:
        adjs.append((torch.from_numpy(g["adj%d_edge_index" % i]), torch.from_numpy(g["adj%d_e_id" % i]),
                     tuple(int(v) for v in g["adj%d_size" % i])))
        i += 1
    return Config(all=Config(x=torch.from_numpy(g["x"]), edge_attr=torch.from_numpy(g["edge_attr"])),
                  batch_n_id=torch.from_numpy(g["n_id"]), batch_adjs=adjs)
