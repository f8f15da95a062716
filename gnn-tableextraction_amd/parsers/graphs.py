"""CLI flag schema of the train/predict entry points.  Same flag names and defaults as the
reference's simple_parsing dataclasses (src/parsers/graphs.py:5-110); plain argparse here
(simple_parsing is not a dependency).  Merge rule as in graphs/utils.py:146-177: a flag given on
the command line overrides the YAML / default value, an omitted flag leaves it alone."""
import argparse

from ..utils.config import AttrDict, default_config


def _bool(v):
    return str(v).lower() in ("1", "true", "yes", "y")


FLAGS = {
    # section, name, type
    "PREPROCESS": [("mode", str), ("features", str), ("edge_features", _bool), ("bidirectional", _bool),
                   ("padding", _bool), ("k", int), ("max_dist", int), ("range_island", int), ("seed", int)],
    "TRAINING": [("num_graphs", int), ("batch_size", int), ("n_layers", int), ("dropout", float), ("lr", float),
                 ("weight_decay", float), ("n_epochs", int), ("es_patience", int), ("gpu", int),
                 ("mode_params", str), ("class_weights", _bool), ("class_weights_method", str)],
    "MODES.fixed": [("h_layer_dim", int)],
    "MODES.scaled": [("params_no", int)],
    "GENERAL": [("from_checkpoint", _bool), ("output_dir", str)],
}


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser("model_train")
    for section, flags in FLAGS.items():
        for name, typ in flags:
            if name == "features":
                ap.add_argument("--features", nargs="+", default=None)
            else:
                ap.add_argument(f"--{name}", type=typ, default=None)
    return ap


def parse_args_ModelTrain(base: AttrDict = None, argv=None) -> AttrDict:
    cfg = base if base is not None else default_config()
    ns = build_parser().parse_args(argv)
    for section, flags in FLAGS.items():
        node = cfg
        for part in section.split("."):
            node = node[part]
        for name, _ in flags:
            v = getattr(ns, name)
            if v is not None:
                node[name] = v
    t, p = cfg.TRAINING, cfg.PREPROCESS
    assert p.mode in ("knn", "visibility"), "mode must be knn or visibility"
    assert set(p.features) <= {"BBOX", "REPR", "SPACY", "SCIBERT"}, "unknown feature set"
    assert t.mode_params in ("fixed", "scaled"), "mode_params must be fixed or scaled"
    return cfg
