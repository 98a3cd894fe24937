"""Attribute-dict config (`clf`) with the keys the hot path reads.

The reference parses its YAML into a ``munch.Munch`` (run.py:291) and mutates a ``clf.temp``
scratch namespace (run.py:294-297).  ``munch`` is not a dependency here; ``Config`` gives the
same attribute access.  The hot path reads only: model.convs / edge_convs / decoder /
normalization, training.loss, regularization.cell_type / edge_type, temp.device,
temp.num_node_features / num_edge_features (SURVEY.md section 5 "Config / flags").
"""
from __future__ import annotations


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return Config({k: Config.wrap(v) for k, v in o.items()})
        if isinstance(o, (list, tuple)):
            return [Config.wrap(v) for v in o]
        return o


def load_config(path: str) -> Config:
    """YAML -> Config, as run.py:291 does with Munch.fromYAML."""
    import yaml

    with open(path) as f:
        clf = Config.wrap(yaml.safe_load(f))
    clf.temp = Config()
    return clf


def reconbench_pretrained(device: str = "cuda:0", convs=(64, 128, 128, 128)) -> Config:
    """The hot-path subset of configs/pretrained/reconbench.yaml (:19-20, :45, :53-61, :66-70) with
    the feature counts of the shipped checkpoint (28 node / 20 edge features)."""
    return Config.wrap(dict(
        features=dict(scaling="s", normalization_range=[0, 1],
                      node_features=["shape", "vertex", "facet", "count", "min", "max", "sum", "first", "second"],
                      edge_features=["shape", "vertex", "facet", "count", "min", "max", "sum"],
                      node_normalization_feature=None, edge_normalization_feature=None),
        model=dict(type="sage", convs=list(convs), edge_convs=1, decoder=2, normalization="b"),
        training=dict(loss="kl", learning_rate=0.005, adjust_lr_every=24, batch_size=2048),
        inference=dict(batch_size=0, per_layer=1, has_label=1),
        graph=dict(num_hops=4, additional_num_hops=1, clique_sizes=[-1], self_loops=0),
        regularization=dict(cell_type="vol", cell_norm=None, edge_type=None, edge_epoch=None, edge_weight=0.4),
        temp=dict(device=device, num_node_features=28, num_edge_features=20),
    ))
