"""gte_sage_smallk_bwd alone at the F0 = 13 step's shape; GTE_LIB_PATH selects an ablation build (-DSKB_ABL=n)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib, torch
ops = importlib.import_module("gnn-tableextraction_amd.ops")
dev = torch.device("cuda:0")
torch.manual_seed(0)
n, k, f = int(sys.argv[1]) if len(sys.argv) > 1 else 24437, 13, 256
x, ahn = torch.randn(n, k, device=dev), torch.randn(n, k, device=dev)
W, b = torch.randn(f, 2 * k, device=dev) / 5, torch.randn(f, device=dev)
g, be = torch.ones(f, device=dev), torch.zeros(f, device=dev)
dy = torch.randn(n, f, device=dev)
stats = torch.cat([torch.zeros(n, device=dev), torch.ones(n, device=dev)])
gW, gb, gg, gbe = torch.empty(f, 2 * k, device=dev), torch.empty(f, device=dev), torch.empty(f, device=dev), torch.empty(f, device=dev)
fn = lambda: ops.sage_smallk_bwd(dy, x, ahn, W, b, g, be, stats, True, gW, gb, gg, gbe)
for _ in range(300): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): fn()
e1.record(); torch.cuda.synchronize()
print(f"[{os.path.basename(os.environ.get('GTE_LIB_PATH', 'default'))}] n={n}: {e0.elapsed_time(e1) * 1e3 / 200:7.1f} us per call (kernel + 4 fold launches)", flush=True)
