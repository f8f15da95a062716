"""cProfile of model_train.train over 20 epochs of 57 steps (6 000 pages, F0 = 831, hidden 256): what an epoch costs besides its steps."""
import cProfile, io, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gnn_tableextraction_amd.components.graphs.loader import PrebuiltPages
from gnn_tableextraction_amd.models import model_train
from gnn_tableextraction_amd.parsers.graphs import parse_args_ModelTrain
import bench
pages = bench.make_pages_parallel(int(sys.argv[1]) if len(sys.argv) > 1 else 6000, 831, 0, min(32, os.cpu_count() or 1))
data = PrebuiltPages(pages)
with tempfile.TemporaryDirectory() as out:
    cfg = parse_args_ModelTrain(argv=["--mode=knn", "--features", "BBOX", "REPR", "SCIBERT", "--n_layers=3", "--mode_params=fixed",
                                      "--h_layer_dim=256", "--batch_size=100", "--n_epochs=20", "--lr=0.01", "--output_dir", out])
    pr = cProfile.Profile()
    pr.enable()
    model_train.train(data, cfg)
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "cumulative").print_stats(60)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:90]))
