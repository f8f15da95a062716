"""cfg3 (GAT bf16) step alone, for rocprofv3 --kernel-trace --stats."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import gnn_tableextraction_amd as gte
args = argparse.Namespace()
print(bench.cfg3_probe(args, gte, torch.device("cuda", 0)))
