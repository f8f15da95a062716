"""nodes/sec of the GcnSAGE train step (fwd + loss + bwd + Adam) on synthetic PubLayNet-style page
graphs -- the metric of BASELINE.json -- on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] / SURVEY 8(d) cfg2): batches of 100 page graphs, 3-layer
GraphSAGE-GCN, F0 = 831 (BBOX+REPR+SCIBERT), hidden 256, 9 classes, fp32, Adam(lr 0.01, wd 5e-4),
unweighted cross-entropy.  Every rank owns DISTINCT pages (weak scaling); the only collective is
one RCCL all-reduce of the flat gradient per step.  The batched graphs (CSR, features, labels) are
resident in HBM before the timed region; a "step" = forward + loss + backward + optimiser on one
resident batch.  One JSON line is printed by rank 0.

Extra objects in the JSON line:
  roofline      dominant kernel (fp32 MFMA forward GEMM), live HIP-event timing over the timed region
  gather        the aggregation kernel on BASELINE cfg4 (1 M nodes, deg 12, F = 512): HBM GB/s
  cpu_baseline  the CPU oracle (oracle/gcnsage_cpu.py: torch-CPU + OpenMP CSR SpMM) on the same
                first batch, on this box's host cores ("port"; baseline only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # fp32 matrix peak (dense)


def pmc_traffic():
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN/pmc_traffic.json: how they
    were collected is in that file).  PMC counters cannot be read from inside this process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))
    if not files:
        return {}, None
    return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)        # >= 50 timed steps after 10 warm-up
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--in-feats", type=int, default=831)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--pages", type=int, default=100, help="page graphs per batch per GPU")
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches cycled through")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather-probe", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the F0=13 (BBOX features only) variant of cfg2")
    ap.add_argument("--gather-nodes", type=int, default=1_000_000)
    ap.add_argument("--val-graph", type=int, default=0, metavar="PAGES",
                    help="also time the forward-only pass over PAGES pages batched into one graph (cfg2 'val graph': 2000)")
    return ap.parse_args()


def build_batches(S, gte, args, rank, dev):
    batches = []
    for b in range(args.batches):
        first = (rank * args.batches + b) * args.pages
        pages = S.make_pages(args.pages, in_feats=args.in_feats, first_id=first)
        src, dst, w, feat, label, off = S.concat_pages(pages)
        g = gte.PageGraph(src, dst, int(off[-1]), device=dev)
        g.ndata["feat"] = torch.from_numpy(feat).to(dev)
        g.edata["feat"] = torch.from_numpy(w).to(dev)
        g.batch_num_nodes_ = [p.num_nodes for p in pages]
        batches.append((g, torch.from_numpy(label).to(dev), (src, dst, w, feat, label, off)))
    return batches


def usable_cores() -> int:
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota).  The GPU box
    shows 256 hardware threads but its cgroup grants a quota (cpu.max); oversubscribing it stalls."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(args, host_batch, state):
    from oracle import gcnsage_cpu as oc
    src, dst, w, feat, label, off = host_batch
    cores = usable_cores()
    torch.set_num_threads(cores)
    oc.set_omp_threads(cores)
    tr = oc.OracleTrainer(state, lr=0.01, weight_decay=5e-4)
    og = oc.OracleGraph(src, dst, int(off[-1]), w)
    x, y = torch.from_numpy(feat), torch.from_numpy(label)
    for _ in range(2):
        tr.step(og, x, y)
    times = []
    t_all = time.time()
    while len(times) < 5 or (time.time() - t_all < 10 and len(times) < 200):      # ~10 s of CPU work
        t0 = time.time()
        tr.step(og, x, y)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {"value": int(off[-1]) / med, "unit": "nodes/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} train steps (median) of the CPU oracle on batch 0 ({int(off[-1])} nodes, "
                      f"{len(src)} edges, F0={args.in_feats}); aggregation = "
                      f"{'OpenMP CSR SpMM' if oc.omp_available() else 'torch.sparse_csr'}, "
                      f"dense ops = torch CPU ({torch.get_num_threads()} threads)",
            "ms_per_step": med * 1e3}


def val_graph_probe(args, gte, S, model, dev):
    """SURVEY 8(d) cfg2, "val graph" case (model_train.py:349-370): every page of a validation set batched into ONE graph,
    forward only (no_grad, eval mode) -- the reference evaluates on the single giant val_graph after each epoch.
    Opt-in (--val-graph PAGES): building the synthetic pages on the host takes longer than the whole default bench."""
    import time as _t
    t0 = _t.perf_counter()
    pages = S.make_pages(args.val_graph, in_feats=args.in_feats, first_id=10_000_000)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    n = int(off[-1])
    g = gte.PageGraph(src, dst, n, device=dev)
    g.ndata["feat"] = torch.from_numpy(feat).to(dev)
    g.edata["feat"] = torch.from_numpy(w).to(dev)
    build_s = _t.perf_counter() - t0
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            logits = model(g)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        s.record()
        for _ in range(reps):
            logits = model(g)
        e.record()
        torch.cuda.synchronize()
    model.train(was_training)
    ms = s.elapsed_time(e) / reps
    return {"workload": f"val graph: {args.val_graph} pages in one graph, forward only (eval, no_grad), F0={args.in_feats}",
            "nodes": n, "edges": int(len(src)), "ms_per_forward": ms, "nodes_per_s": n / (ms * 1e-3),
            "host_build_s": build_s, "logits_finite": bool(torch.isfinite(logits).all())}


def secondary_probe(args, gte, S, dev):
    """SURVEY 8(d) cfg2, secondary width: the same step with BBOX features only (F0 = 13; 16 of the reference's 96 ablation
    runs).  Same pages, same model sizes otherwise; resident batches, HIP-graph replay; a short run of its own."""
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    a = argparse.Namespace(**vars(args))
    a.in_feats, a.batches = 13, 2
    batches = build_batches(S, gte, a, 0, dev)
    torch.manual_seed(42)
    model = gte.GcnSAGE(13, args.hidden, 9, args.layers, torch.nn.functional.relu, 0).to(dev)
    trainer = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
    replays = [trainer.capture(g, y) for g, y, _ in batches]
    for i in range(8):
        replays[i % 2]()
    torch.cuda.synchronize()
    steps = 40
    t0 = time.perf_counter()
    for i in range(steps):
        out3 = replays[i % 2]()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nodes = sum(batches[i % 2][0].num_nodes() for i in range(steps))
    dims = [13] + [args.hidden] * (args.layers - 1) + [9]
    flops_node = sum(2.0 * 2 * dims[l] * dims[l + 1] * (3 if l > 0 else 2) for l in range(args.layers))
    return {"workload": f"cfg2 secondary: F0=13 (BBOX features only), {args.pages} pages per step, hidden={args.hidden}",
            "value": nodes / el, "unit": "nodes/s", "steps": steps, "ms_per_step": el / steps * 1e3,
            "final_loss": float(out3[0]), "mfma_bound_nodes_per_s": MFMA_F32_PEAK_TF * 1e12 / flops_node,
            "frac_of_mfma_bound": nodes / el / (MFMA_F32_PEAK_TF * 1e12 / flops_node)}


def gather_probe(args, gte, S, dev):
    """Aggregation kernel alone on BASELINE cfg4 (HBM-bandwidth stress)."""
    from gnn_tableextraction_amd import ops
    n, k, f = args.gather_nodes, 12, 512
    src, dst, w = S.make_knn_stress_graph(n, k)
    indptr, indices, perm, wout = ops.coo_to_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n,
                                                 torch.from_numpy(w).to(dev))
    x = torch.randn(n, f, device=dev)
    out = torch.empty_like(x)
    plan = ops.build_tile_plan(indptr, indices, n)          # graph structure, built once per graph

    def timed(tiles):
        for _ in range(3):
            ops.spmm_csr(indptr, indices, wout, x, n, mean=True, out=out, tiles=tiles, force_tiled=tiles is not None)
        torch.cuda.synchronize()
        reps = 10
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for s, e in evs:
            s.record()
            ops.spmm_csr(indptr, indices, wout, x, n, mean=True, out=out, tiles=tiles, force_tiled=tiles is not None)
            e.record()
        torch.cuda.synchronize()
        return float(np.mean([s.elapsed_time(e) for s, e in evs]))

    ms_plain = timed(None)
    ms = timed(plan)
    # yardstick: a plain device copy of the same feature matrix into the same output (reads X once, writes out once --
    # the compulsory traffic of the aggregation minus the edge list)
    for _ in range(3):
        out.copy_(x)
    torch.cuda.synchronize()
    cs, ce = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cs.record()
    for _ in range(10):
        out.copy_(x)
    ce.record()
    torch.cuda.synchronize()
    copy_ms = cs.elapsed_time(ce) / 10
    # (iii) of SURVEY 8(d) cfg4: one full GcnSAGELayer(512 -> 512) forward + backward on the same graph
    layer_ms = None
    if n == 1_000_000:
        torch.manual_seed(0)
        layer = gte.GcnSAGELayer(f, f, torch.nn.functional.relu, 0).to(dev)
        g = gte.PageGraph(src, dst, n, device=dev)
        g.edata["feat"] = torch.from_numpy(w).to(dev)
        h = x.clone().requires_grad_(True)
        up = torch.randn(n, f, device=dev)
        def fb():
            y = layer(g, h)
            y.backward(up)
            layer.zero_grad(set_to_none=True); h.grad = None
        for _ in range(2):
            fb()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fb()
        torch.cuda.synchronize()
        layer_ms = (time.perf_counter() - t0) / 5 * 1e3
        del layer, g, h, up
    alg_bytes = 2.0 * n * f * 4 + 8.0 * n * k + 4.0 * (n + 1)      # 2*F*s + 8*d + 4 per node
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"workload": f"cfg4: 1 graph, {n} nodes, in-degree {k}, F={f} fp32, k-NN of 2-D points in Morton order",
            "kernel": "spmm_tiled_full_kernel (LDS-staged distinct sources, two chunks in flight)", "plain_kernel_ms": ms_plain,
            "plain_kernel_GBs": alg_bytes / (ms_plain * 1e-3) / 1e9, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "ms_per_pass": ms, "algorithmic_bytes": alg_bytes,
            "nodes_per_s_per_pass": n / (ms * 1e-3),
            "device_copy_same_matrix_ms": copy_ms, "device_copy_GBs": 2.0 * n * f * 4 / (copy_ms * 1e-3) / 1e9,
            "full_layer_512_fwd_bwd_ms": layer_ms,
            "full_layer_nodes_per_s": (n / (layer_ms * 1e-3)) if layer_ms else None,
            "traffic": pmc_traffic()[0].get("gather_cfg4_tiled_bytes_per_launch") if n == 1_000_000 else None,
            "traffic_source": pmc_traffic()[1]}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GTE_BENCH_FORCE_DIST=1 (test hook): run the data-parallel code path -- RCCL process group, flat-gradient all-reduces,
    # HIP graph + eager collective + Adam -- even with one rank, so a 1-GPU box exercises RCCL next to graph capture.
    distributed = world > 1 or os.environ.get("GTE_BENCH_FORCE_DIST", "0") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    # Test hook for 1-GPU boxes (GTE_BENCH_SHARE_GPU=1): every rank uses cuda:0 and the ranks talk over gloo (RCCL refuses two
    # ranks on one device) -- exercises the N > 1 code path of this file, not a measurement.
    share_gpu = os.environ.get("GTE_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import ops
    from gnn_tableextraction_amd.data import synthetic as S
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep

    batches = build_batches(S, gte, args, rank, dev)
    torch.manual_seed(42)
    model = gte.GcnSAGE(args.in_feats, args.hidden, 9, args.layers, torch.nn.functional.relu, 0)
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    trainer = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4, distributed=distributed)

    # global node count of step i (every rank can compute it: page sizes are seeded metadata)
    local_nodes = [b[0].num_nodes() for b in batches]
    if distributed:
        t = torch.tensor(local_nodes, dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        global_nodes = t.cpu().tolist()
    else:
        global_nodes = local_nodes

    # one HIP graph per resident batch (forward + loss + backward); all-reduce and Adam follow eagerly
    replays = None
    if not args.no_graph:
        replays = [trainer.capture(g, y, n_global=global_nodes[i]) for i, (g, y, _) in enumerate(batches)]

    def run(i, eager=False):
        j = i % len(batches)
        if replays is not None and not eager:
            return replays[j]()
        g, y, _ = batches[j]
        return trainer.step(g, y, n_global=global_nodes[j])

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        run(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out3 = run(i)
    barrier()
    elapsed = time.perf_counter() - t0
    final_loss = float(out3[0])
    # per-kernel HIP-event timing: the SAME steps launched eagerly (events cannot sit between the nodes of
    # a graph replay), on the launch stream, right after the timed region
    for i in range(len(batches)):                # untimed: the eager launch pattern settles (clocks, caches)
        run(i, eager=True)
    ops.enable_kernel_timers(True)
    for i in range(len(batches) * 2):
        run(i, eager=True)
    kt = ops.kernel_timer_report()
    ops.enable_kernel_timers(False)

    nodes_local = sum(local_nodes[i % len(batches)] for i in range(args.steps))
    stat = torch.tensor([elapsed, float(nodes_local)], dtype=torch.float64, device=dev)
    if distributed:
        tmax = stat.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)
        elapsed, nodes_total = float(tmax[0]), float(stat[1])
    else:
        nodes_total = float(nodes_local)

    if rank == 0:
        n_launch, ms, flops = kt.get("gemm_nt", (0, 0.0, 0.0))
        tf = (flops / (ms * 1e-3) / 1e12) if ms > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": "gemm_f32_mfma_kernel<NT> (layer transforms, forward)",
                    "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                    "launches": n_launch, "avg_launch_ms": ms / max(n_launch, 1),
                    "algorithmic_flops_per_launch": flops / max(n_launch, 1),
                    "traffic": pmc_traffic()[0].get("gemm_nt_bytes_per_launch") if args.in_feats == 831 else None,
                    "traffic_source": pmc_traffic()[1]}
        per_kernel = {}
        for tag, (n, tms, work) in kt.items():
            unit = "GB/s" if tag == "spmm_csr" else "TFLOP/s"
            rate = work / (tms * 1e-3) / (1e9 if tag == "spmm_csr" else 1e12) if tms > 0 else 0.0
            per_kernel[tag] = {"launches": n, "total_ms": tms, "avg_ms": tms / max(n, 1), "rate": rate, "unit": unit}
        line = {
            "metric": "nodes/sec (fwd+bwd node classification) on PubLayNet page graphs",
            "value": nodes_total / elapsed, "unit": "nodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg2: {args.pages} synthetic PubLayNet-style page graphs per GPU per step "
                                   f"(~{int(np.mean(local_nodes))} nodes, k-NN k=5 bidirected), GcnSAGE "
                                   f"{args.layers} layers F0={args.in_feats} hidden={args.hidden} classes=9, "
                                   f"CE + Adam(lr 0.01, wd 5e-4); batches resident in HBM; "
                                   f"{'HIP-graph replay per batch' if replays is not None else 'eager launches'}",
                       "pages_per_gpu_per_step": args.pages, "global_pages_per_step": args.pages * world,
                       "nodes_per_step_per_gpu": int(np.mean(local_nodes)), "parallelism": f"dp{world}"},
            "final_loss": final_loss, "roofline": roofline, "kernels": per_kernel,
        }
        # whole-step bounds of SURVEY 8(d): per node, layer l (F_l -> F_{l+1}, mean in-degree d), training mode, no recomputation
        dims = [args.in_feats] + [args.hidden] * (args.layers - 1) + [9]
        deg = float(np.mean([b[0].in_csr().indices.numel() / max(b[0].num_nodes(), 1) for b in batches]))
        flops_node = sum(2.0 * 2 * dims[l] * dims[l + 1] * (3 if l > 0 else 2) for l in range(args.layers))
        bytes_node = sum(4.0 * (2 * dims[l] + dims[l + 1]) + 8 * deg + 4 +                       # forward
                         4.0 * (dims[l + 1] + 2 * dims[l] + (dims[l] if l > 0 else 0)) + 8 * deg + 4  # backward
                         for l in range(args.layers))
        mfma_bound = MFMA_F32_PEAK_TF * 1e12 / flops_node * world
        hbm_bound = HBM_PEAK_GBS * 1e9 / bytes_node * world
        line["step_roofline"] = {"flops_per_node": flops_node, "bytes_per_node": bytes_node, "mean_in_degree": deg,
                                 "mfma_bound_nodes_per_s": mfma_bound, "hbm_bound_nodes_per_s": hbm_bound,
                                 "bound": "mfma" if mfma_bound < hbm_bound else "hbm",
                                 "frac": line["value"] / min(mfma_bound, hbm_bound)}
        if world == 1 and not args.no_gather_probe:
            line["gather"] = gather_probe(args, gte, S, dev)
        if world == 1 and not distributed and not args.no_secondary and args.in_feats != 13:
            line["secondary"] = secondary_probe(args, gte, S, dev)
        if world == 1 and args.val_graph > 0:
            line["val_graph"] = val_graph_probe(args, gte, S, model, dev)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args, batches[0][2], state0)
            line["cpu_baseline"] = cb
            line["gpu_over_cpu"] = line["value"] / cb["value"]
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
