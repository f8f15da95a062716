"""bench.residency_probe in a fresh process (is the probe slower inside the full bench run?)"""
import os, sys, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models import loop
args = argparse.Namespace(in_feats=831, hidden=256, layers=3, pages=100)
pages = S.make_pages(3000, in_feats=831)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = bench.residency_probe(args, gte, dev, pages, loop)
print(json.dumps(out["windowed"], indent=1), out["all_resident"])
