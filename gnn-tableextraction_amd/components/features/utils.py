"""Shape helpers of the train entry point (reference src/components/features/utils.py:71-101)."""
import math

FEATURE_WIDTHS = {'BBOX': 13, 'REPR': 50, 'SPACY': 300, 'SCIBERT': 768}


def get_in_feats_(config):
    """Input width F0 = sum of the chosen embedders' widths; with ``padding`` always
    BBOX+REPR+SCIBERT = 831."""
    names = ['BBOX', 'REPR', 'SCIBERT'] if config.PREPROCESS.padding else config.PREPROCESS.features
    return sum(FEATURE_WIDTHS[n] for n in names)


def calculate_hidden(input_dim, classes_no, params_no, layer_no):
    """Hidden width h with (L-1) h^2 + (C + F0) h = P: the larger root (callers take int())."""
    a, b = layer_no - 1, classes_no + input_dim
    root = math.sqrt(b * b + 4 * a * params_no)
    return max((-b - root) / (2 * a), (-b + root) / (2 * a))
