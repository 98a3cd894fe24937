"""CPU tests of bench.py's own machinery (the pieces that decide what a bench line claims): the randomly relabelled scene is the same graph,
the logit check flags what it should, the algorithmic byte counts are SURVEY 8d's."""
import numpy as np
import torch

import bench
from dgnn_amd.config import Config
from dgnn_amd.synthetic import check_four_regular
from helpers import oracle_static


def test_relabelled_scene_is_the_same_graph_with_other_cell_numbers():
    adj, _, x, ea = bench.make_scene(300, 0)
    adj2, x2, ea2 = bench.relabelled_scene(adj, x, ea)
    n = adj.shape[0] // 4
    assert check_four_regular(adj2) and adj2.shape == adj.shape and not np.array_equal(adj, adj2)
    perm = np.random.default_rng(7).permutation(n)
    net = oracle_static()
    with torch.no_grad():
        a = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(adj.T.astype(np.int64))))
        b = net.inference_layer(Config(x=x2, edge_attr=ea2, edge_index=torch.from_numpy(adj2.T.astype(np.int64))))
    assert (b[torch.from_numpy(perm)] - a).abs().max().item() <= 2e-5      # same cells, same neighbours, another summation order per cell
    # locality is gone: the generator's order keeps most neighbours close, the relabelled one does not
    d1 = np.abs(adj[:, 0].astype(np.int64) - adj[:, 1]).astype(np.float64)
    d2 = np.abs(adj2[:, 0].astype(np.int64) - adj2[:, 1]).astype(np.float64)
    assert np.median(d2) > 10 * np.median(d1)


def test_logits_check_accepts_rounding_noise_and_flags_real_errors():
    g = torch.Generator().manual_seed(0)
    ref = torch.randn(5000, 2, generator=g) * 3
    ok = bench.logits_check(ref + 5e-6 * torch.randn(5000, 2, generator=g), ref, bf16=False)
    assert ok["ok"] and ok["argmax_flips_above_margin"] == 0 and ok["max_abs_err"] < 1e-4
    bad = ref.clone()
    bad[17, 0] += 2e-3
    assert not bench.logits_check(bad, ref, bf16=False)["ok"]
    swapped = ref.clone()
    k = int((ref[:, 0] - ref[:, 1]).abs().argmax())
    swapped[k] = ref[k].flip(0)                      # a label flip far above the margin
    r = bench.logits_check(swapped, ref, bf16=False)
    assert not r["ok"] and r["argmax_flips_above_margin"] >= 1
    # bf16 storage: the stated bound is relative to max(1, |logit| / 8)
    noisy = ref + 3e-3 * torch.randn(5000, 2, generator=g).clamp(-2, 2)
    assert bench.logits_check(noisy, ref, bf16=True)["ok"] and not bench.logits_check(ref + 0.3, ref, bf16=True)["ok"]


def test_algorithmic_bytes_are_the_survey_figures():
    assert [bench.layer_bytes(a, b) for a, b in ((28, 64), (64, 128), (128, 128))] == [704, 1104, 1360]
    assert bench.path_bytes(28, (64, 128, 128, 128)) == 5048 and bench.path_bytes(28, (64, 128, 128, 128), 2) == 3200
    assert bench.layer_flops(128, 128) == 87040
