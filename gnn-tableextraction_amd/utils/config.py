"""Attribute-style config dict (the reference wraps its YAML in ``attrdict.AttrDict``:
src/models/model_train.py:462-464) and the run-name builder (graphs/utils.py:287-306)."""


class AttrDict(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, AttrDict):
                self[k] = AttrDict(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


def default_config(**training) -> AttrDict:
    """The keys of configs/graph/empty.yaml that the train loop reads, with the reference's defaults
    (src/parsers/graphs.py:21-80)."""
    cfg = AttrDict({
        "GENERAL": {"from_checkpoint": False, "converted": True, "output_dir": "output"},
        "PREPROCESS": {"mode": "knn", "features": ["BBOX"], "edge_features": True, "bidirectional": True,
                       "padding": False, "k": 5, "max_dist": 500, "range_island": 2, "seed": 42},
        "TRAINING": {"num_graphs": None, "batch_size": 100, "n_layers": 3, "dropout": 0, "lr": 0.01,
                     "weight_decay": 5e-4, "n_epochs": 2000, "es_patience": 50, "gpu": 0,
                     "mode_params": "fixed", "class_weights": False, "class_weights_method": "default",
                     "h_layer_dim": None},
        "MODES": {"fixed": {"h_layer_dim": 1000}, "scaled": {"params_no": 100000}},
    })
    cfg.TRAINING.update(training)
    return cfg


def logs_from_config(config) -> str:
    """Run name: {num_graphs|all}[-cw]-{mode}-nfeat_{..}-[efeat-][dibi-]bt_{B}-nlay_{L}-rhop_{r}-pmode_{..}-..."""
    t, p = config.TRAINING, config.PREPROCESS
    name = str(t.num_graphs) + '-' if t.num_graphs is not None else 'all'
    if t.class_weights:
        name += 'cw-'
    name += f"{p.mode}-nfeat_{'_'.join(p.features)}-"
    if p.edge_features:
        name += 'efeat-'
    if p.bidirectional:
        name += 'dibi-'
    name += f"bt_{t.batch_size}-nlay_{t.n_layers}-rhop_{p.range_island}-pmode_{t.mode_params}-"
    if t.mode_params == 'fixed':
        name += f"hdim_{config.MODES.fixed.h_layer_dim}"
    elif t.mode_params == 'scaled':
        name += f"pno_{config.MODES.scaled.params_no}"
    return name
