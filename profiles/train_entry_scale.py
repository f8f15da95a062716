"""The drop-in entry point end to end: model_train.train(data, config) on a synthetic PubLayNet-style set at the reference's own
flags (run_multiple_train.sh: --features BBOX REPR SCIBERT = 831 inputs, --h_layer_dim=1000 | --mode_params=scaled), with and
without an HBM budget for the training set.  Prints one JSON line per run: seconds per epoch (train steps + validation), the
implied nodes/s of the train steps, final metrics.
    python profiles/train_entry_scale.py [pages=6000] [epochs=3] [budget_GB=0 (all resident) | 4 ...]"""
import json, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnn_tableextraction_amd.components.graphs.loader import PrebuiltPages
from gnn_tableextraction_amd.models import model_train
from gnn_tableextraction_amd.parsers.graphs import parse_args_ModelTrain
import bench

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
n_epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hidden = sys.argv[4] if len(sys.argv) > 4 else "1000"
t0 = time.perf_counter()
pages = bench.make_pages_parallel(n_pages, 831, 0, min(32, os.cpu_count() or 1))       # (before anything initialises the GPU)
data = PrebuiltPages(pages)
gen_s = time.perf_counter() - t0
if budget > 0:
    os.environ["GTE_RESIDENT_BUDGET_GB"] = str(budget)
with tempfile.TemporaryDirectory() as out:
    mode = ["--mode_params=scaled", "--params_no=100000"] if hidden == "scaled" else ["--mode_params=fixed", f"--h_layer_dim={hidden}"]
    cfg = parse_args_ModelTrain(argv=["--mode=knn", "--features", "BBOX", "REPR", "SCIBERT", "--n_layers=3", *mode, "--batch_size=100",
                                      f"--n_epochs={n_epochs}", "--lr=0.01", "--output_dir", out])
    torch.cuda.synchronize() if torch.cuda.is_initialized() else None
    t1 = time.perf_counter()
    metrics = model_train.train(data, cfg)
    torch.cuda.synchronize()
    el = time.perf_counter() - t1
nodes = int(sum(p.num_nodes for p in pages))
print(json.dumps({"pages": n_pages, "nodes": nodes, "epochs": n_epochs, "hidden": hidden, "budget_GB": budget,
                  "page_generation_s": gen_s, "train_call_s": el, "s_per_epoch_incl_setup": el / n_epochs,
                  "val_loss": float(metrics.val.loss), "train_loss": float(metrics.train.loss)}))
