"""nodes/sec of the GcnSAGE train step (fwd + loss + bwd + Adam) on synthetic PubLayNet-style page
graphs -- the metric of BASELINE.json -- on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a torchrun environment: this process starts N fresh ranks itself (torch.distributed.run,
      one per GPU, RCCL over xGMI) BEFORE it touches any GPU, relays rank 0's JSON line and exits with their code.
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (the driver's form)

Workload (BASELINE.json configs[1] / SURVEY 8(d) cfg2): batches of 100 page graphs, 3-layer GraphSAGE-GCN, F0 = 831
(BBOX+REPR+SCIBERT), hidden 256, 9 classes, fp32, Adam(lr 0.01, wd 5e-4), unweighted cross-entropy.

What is timed is the loop ``models.model_train.train`` runs (``models/loop.py: run_steps``, the same function): every
step takes a DIFFERENT set of 100 pages out of the rank's resident dataset (shuffled epochs of
``distributed.plan_epoch``, as model_train.py:279-283), the batched graph is assembled on the device on a side stream
while the previous step runs, then forward + loss + backward + (all-reduce) + Adam.  The pages (features, labels, CSRs)
are resident in HBM before the timed region.  Every rank owns DISTINCT pages (weak scaling); the only collective is one
RCCL all-reduce of the flat gradient per step.  One JSON line is printed by rank 0.

Extra objects in the JSON line:
  roofline      dominant kernel (the forward transform GEMMs: planes GEMM on the bf16 matrix pipe by default), live HIP-event
                timing of the same steps
  long_run      the same loop for >= 1 s of device time (after a few warm-up steps of its own), run BEFORE the W warm-up + K timed steps (the driver's 20-step
                region is ~15 ms; measured right after an idle GPU woke up it reads ~7 % low: the chip has not reached its
                sustained clocks -- a training run is at them)
  replay        secondary: HIP-graph replay of pre-captured resident batches (round 1's headline mode)
  gather        the aggregation kernel on BASELINE cfg4 (1 M nodes, deg 12, F = 512): HBM GB/s
  cfg3          BASELINE configs[2]: 4-head GAT, bf16 MFMA projection + bf16 gathers (no reference counterpart: parity unpinned)
  val_graph     cfg2 "val graph" case: every page of a validation set in ONE graph, forward only (model_train.py:349-353)
  inference     model_predict.py:130-154: one forward per page (eager / one HIP-graph launch per page) and 100 pages per forward
  cpu_baseline  the CPU oracle (oracle/gcnsage_cpu.py: torch-CPU + OpenMP CSR SpMM) on one batch, on this box's
                host cores ("port"; baseline only)
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
MFMA_BF16_PEAK_TF = 2500.0   # bf16 matrix peak (dense; the headline figures with 2:1 sparsity are not used)


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
    ap.add_argument("--resident-pages", type=int, default=1200,
                    help="pages of the resident dataset per GPU; an epoch is resident-pages / pages steps")
    ap.add_argument("--long-run-seconds", type=float, default=1.0, help="device time of the secondary long run (0: skip)")
    ap.add_argument("--no-replay", action="store_true", help="skip the HIP-graph replay secondary")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather-probe", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="skip the inference probe (one forward per page / batched)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the F0=13 (BBOX features only) variant of cfg2")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the GAT bf16 probe (BASELINE configs[2])")
    ap.add_argument("--no-shapes", action="store_true", help="skip the reference's own run shapes (hidden 1000 / scaled hidden)")
    ap.add_argument("--no-size-sweep", action="store_true", help="skip the pages-per-step sweep of the headline configuration")
    ap.add_argument("--no-residency", action="store_true", help="skip the host-resident (windowed) training-set probe")
    ap.add_argument("--no-uncached", action="store_true", help="skip the loop without the cached input aggregate (`uncached`)")
    ap.add_argument("--no-dist-probe", action="store_true", help="skip the one-rank data-parallel step (RCCL group of one) against the plain step")
    ap.add_argument("--gemm-mode", choices=["f32", "split_bf16"], default=None,
                    help="arithmetic of the transform GEMMs for the headline loop (default: GTE_GEMM_MODE or f32)")
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="skip the per-kernel HIP-event pass (the launch-by-launch schedule; profiles/sequence_refresh.sh traces the loop alone)")
    ap.add_argument("--no-split-probe", action="store_true",
                    help="skip the secondary run of the same loop in the split-bf16 GEMM mode")
    ap.add_argument("--gather-nodes", type=int, default=1_000_000)
    ap.add_argument("--val-graph", type=int, default=2000, metavar="PAGES",
                    help="forward-only pass over PAGES pages batched into one graph (cfg2 'val graph'; 0: skip)")
    return ap.parse_args()


def maybe_spawn(args):
    """``python bench.py --gpus N`` outside torchrun: start N fresh ranks (one per GPU) and relay them.  Runs before this
    process has made any HIP call -- a process that has initialised the GPU must never be replaced or forked into ranks."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}; they must agree")
        return
    if args.gpus <= 1:
        return
    share = os.environ.get("GTE_BENCH_SHARE_GPU", "0") == "1"
    have = torch.cuda.device_count()             # counting devices does not initialise the GPU
    if have < args.gpus and not share:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s)")
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
    rc = subprocess.call(cmd, env=env)                       # the ranks inherit stdout: rank 0's JSON line is relayed as is
    sys.exit(rc)


def _make_page_job(job):
    from gnn_tableextraction_amd.data import synthetic as S
    pid, in_feats = job
    return S.make_page(pid, in_feats=in_feats)


def make_pages_parallel(n_pages, in_feats, first_id, workers):
    """Synthetic pages on a process pool.  Called BEFORE the GPU is initialised (fork is only safe then)."""
    from gnn_tableextraction_amd.data import synthetic as S
    # under rocprofv3 the forked workers inherit the profiler's preloaded tool and now and then never exit (the parent then
    # sits in wait4 until the run's timeout: it cost two profile runs): generate in-process there
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower()
    if workers <= 1 or n_pages < 64 or under_profiler:
        return S.make_pages(n_pages, in_feats=in_feats, first_id=first_id)
    import multiprocessing as mp
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_make_page_job, [(first_id + i, in_feats) for i in range(n_pages)], chunksize=8)


def to_page_graphs(gte, pages):
    """Host PageGraph per page with the loader's output contract (ndata feat/label, edata feat): loader.py:332-354."""
    out = []
    for p in pages:
        g = gte.PageGraph(p.src, p.dst, p.num_nodes)
        g.ndata["feat"] = torch.from_numpy(p.feat)
        g.ndata["label"] = torch.from_numpy(p.label.astype(np.float32))      # stored as float32 (loader.py:350-354)
        g.edata["feat"] = torch.from_numpy(p.weight)
        out.append(g)
    return out


def epoch_steps(sizes, pages_per_step, seed, first_epoch, n_steps):
    """n_steps page-id lists, epoch by epoch (distributed.plan_epoch: the shuffle + tail drop of model_train.py:279-283),
    as a list of per-epoch lists (the last one may be a partial epoch).  Returns (epochs, next epoch number)."""
    from gnn_tableextraction_amd import distributed as D
    out, e = [], first_epoch
    left = n_steps
    while left > 0:
        plan = [ranks[0] for ranks in D.plan_epoch(sizes, pages_per_step, 1, seed=seed, epoch=e)]
        out.append(plan[:left])
        left -= len(out[-1])
        e += 1
    return out, e


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
            "sample_short": (f"{len(times)} train steps (median) on ONE {int(off[-1])}-node batch of the workload repeated (warm caches; the GPU "
                             f"loop assembles a different batch every step), {time.time() - t_all:.0f} s"),
            "ms_per_step": med * 1e3}


def dist_one_rank_probe(args, gte, dev, resident, pipe, sizes, loop, ep):
    """The data-parallel step with ONE rank through the real RCCL process group against the one-GPU step, same loop, interleaved:
    what the collective's launch and the split of the optimiser launch cost before any wire time (the fixed part of every weak-
    scaling step).  The data-parallel step: fold launch -> all-reduce of the flat gradient -> ONE launch for Adam + the next
    forward's weight images (gte_adam_step_dev_images)."""
    import torch.distributed as dist
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    if dist.is_initialized():
        return None
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env_keep = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    try:
        import datetime
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
        trainers = {}
        for dp in (False, True):
            torch.manual_seed(42)
            m = gte.GcnSAGE(args.in_feats, args.hidden, 9, args.layers, torch.nn.functional.relu, 0).to(dev)
            trainers[dp] = FusedGcnSageStep(m, lr=0.01, weight_decay=5e-4, distributed=dp)
        best = {False: None, True: None}
        for rnd in range(4):
            for dp in (False, True):
                epochs, ep = epoch_steps(sizes, args.pages, 4242, ep, 12 if rnd == 0 else 96)
                counts = [[int(sum(sizes[i] for i in ids)) for ids in plan] for plan in epochs]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                k = 0
                for plan, cnt in zip(epochs, counts):
                    loop.run_steps(trainers[dp], pipe, plan, n_global=cnt if dp else None)
                    k += len(plan)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / k * 1e3
                if rnd > 0 and (best[dp] is None or ms < best[dp]):
                    best[dp] = ms
        return {"plain_ms_per_step": best[False], "dist_1rank_ms_per_step": best[True], "extra_us": (best[True] - best[False]) * 1e3,
                "how": "96 steps of the train loop per arm, three interleaved rounds after a warm-up round, the best of each arm "
                       "(round 5: 48 x 2 read +7 ... +20 us from run to run); "
                       "RCCL process group of one rank"}
    except Exception as e:                                   # (no RCCL on this box / rendezvous refused: the probe is optional)
        return {"error": f"{type(e).__name__}: {e}"[:200]}
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in env_keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def val_graph_probe(args, gte, S, model, dev, pages, eng=None):
    """SURVEY 8(d) cfg2, "val graph" case (model_train.py:246,349-353): every page of a validation set batched into ONE
    graph, forward only (no_grad, eval mode) -- the reference evaluates on the single giant val_graph after each epoch."""
    src, dst, w, feat, label, off = S.concat_pages(pages)
    n = int(off[-1])
    g = gte.PageGraph(src, dst, n, device=dev)
    g.ndata["feat"] = torch.from_numpy(feat).to(dev)
    g.edata["feat"] = torch.from_numpy(w).to(dev)
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
    out = {"workload": f"val graph: {len(pages)} pages in one graph, forward only (eval, no_grad), F0={args.in_feats}",
           "nodes": n, "edges": int(len(src)), "logits_finite": bool(torch.isfinite(logits).all()),
           "module_path": {"ms_per_forward": ms, "nodes_per_s": n / (ms * 1e-3)}}
    # what train() runs every epoch: the engine's forward-only call on the validation graph, whose feature image is made ONCE
    # (the graph is the same every epoch: model_train.py:246 of the reference)
    if eng is not None and eng.attach_feature_image(g):
        for _ in range(3):
            lg = eng.forward_logits(g)
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            lg = eng.forward_logits(g)
        e.record()
        torch.cuda.synchronize()
        ms2 = s.elapsed_time(e) / reps
        out["engine_path"] = {"ms_per_forward": ms2, "nodes_per_s": n / (ms2 * 1e-3),
                              "max_abs_diff_vs_module_path": float((lg - logits).abs().max()),
                              "how": "engine.forward_logits (gte_gcnsage_forward) on the cached feature image"}
        ms = min(ms, ms2)
    out["ms_per_forward"], out["nodes_per_s"] = ms, n / (ms * 1e-3)
    return out


def inference_probe(args, gte, model, dev, pages, trainer=None):
    """SURVEY 8(f) N2 / model_predict.py:130-154.  (i) the reference's shape: ONE forward per page (batch = 1), eager and as
    one HIP-graph launch per page (models.model_predict.PageForwardGraphs: size buckets, the page assembled into the bucket's
    buffers); (ii) batched: 100 pages per forward.  Pages are resident in HBM with their CSRs prepared (the reference rebuilds
    them per forward inside DGL); host-synchronised wall time over all pages.  (iii) batched through the step engine:
    model_predict.predict_resident -- one host call per forward (gte_gcnsage_forward), batches assembled one forward ahead on the
    side stream, predictions written into one device vector."""
    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd.models.model_predict import PageForwardGraphs, predict_resident
    from gnn_tableextraction_amd.models.loop import BatchPipeline
    n_pages = min(len(pages), 400)
    res = G.ResidentPages(to_page_graphs(gte, pages[:n_pages]), dev)
    sizes = res.page_sizes()
    was_training = model.training
    model.eval()
    out = {"workload": f"inference: {n_pages} pages ({int(sum(sizes))} nodes), F0={args.in_feats}, hidden={args.hidden}, forward + argmax"}
    with torch.no_grad():
        def eager_all():
            for pid in range(n_pages):
                model(res.batch([pid])).argmax(dim=1)
        runner = PageForwardGraphs(model, res)
        def graph_all():
            for pid in range(n_pages):
                runner.forward(pid)
        def batched_all():
            for b0 in range(0, n_pages, 100):
                model(res.batch(list(range(b0, min(b0 + 100, n_pages))))).argmax(dim=1)
        for name, fn in (("per_page_eager", eager_all), ("per_page_hip_graph", graph_all), ("batched_100", batched_all)):
            fn()                                                # warm-up (captures the buckets' graphs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            out[name] = {"ms_per_page": el * 1e3 / n_pages, "pages_per_s": n_pages / el, "nodes_per_s": float(sum(sizes)) / el}
        out["hip_graph_buckets"] = sorted(runner._b)
        if trainer is not None and hasattr(trainer, "forward_logits"):
            # LAST: it switches the resident pages to the engine's feature format (a P3 image in the split GEMM mode)
            pipe = BatchPipeline(res)
            engine_all = lambda: predict_resident(trainer, pipe, 100)
            ref = torch.cat([model(res.batch(list(range(b0, min(b0 + 100, n_pages))))).argmax(dim=1) for b0 in range(0, n_pages, 100)])
            got = engine_all()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                engine_all()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / reps
            out["batched_100_engine"] = {"ms_per_page": el * 1e3 / n_pages, "pages_per_s": n_pages / el, "nodes_per_s": float(sum(sizes)) / el,
                                         "predictions_equal_module_path": float((got == ref).float().mean())}
    model.train(was_training)
    return out


def timed_loop(trainer, pipe, epochs, loop):
    """run_steps over the given per-epoch page lists; returns (seconds, nodes, last out3).  Synchronises both ends."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nodes, out3 = 0, None
    for plan in epochs:
        out3 = loop.run_steps(trainer, pipe, plan)
        nodes += sum(pipe.nodes(i) for i in range(len(plan)))
    torch.cuda.synchronize()
    return time.perf_counter() - t0, nodes, out3


def gemm_mode_probe(ops, run, epochs, counts, dev):
    """The SAME train loop in the OTHER arithmetic mode of the transform GEMMs (csrc/gemm_split.h; include/gte.h "GEMM
    arithmetic mode"), plus both modes' error against fp64 on one forward-shaped product -- measured here, every run."""
    cur = ops.get_gemm_mode()
    other = ops.GEMM_F32 if cur == ops.GEMM_SPLIT_BF16 else ops.GEMM_SPLIT_BF16
    names = {ops.GEMM_F32: "f32 (v_mfma_f32_32x32x2_f32)", ops.GEMM_SPLIT_BF16: "split_bf16 (3 exact bf16 pieces, 6 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)"}
    out = {"headline_mode": names[cur]}
    # error of X W^T against fp64, in units of 2^-24 sum_k |x_k w_k| (one fp32 rounding of the product's magnitude)
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(2048, 831, generator=g) * (1 + torch.arange(831) % 7)
    w = torch.randn(512, 831, generator=g) * 0.05
    ref = x.double() @ w.double().t()
    unit = (x.double().abs() @ w.double().abs().t()) * 2.0 ** -24
    xd, wd = x.to(dev), w.to(dev)
    err = {}
    for m in (ops.GEMM_F32, ops.GEMM_SPLIT_BF16):
        ops.set_gemm_mode(m)
        e = ((ops.gemm(xd, wd, trans_b=True).double().cpu() - ref).abs() / unit)
        err["f32" if m == ops.GEMM_F32 else "split_bf16"] = {"max": float(e.max()), "rms": float(e.pow(2).mean().sqrt())}
    out["error_vs_fp64"] = {"unit": "2^-24 * sum_k |x_k w_k|", "shape": "2048 x 831 @ 831 x 512", **err}
    ops.set_gemm_mode(other)
    flat = [(plan, cnt) for plan, cnt in zip(epochs, counts)]
    # warm-up: the first 32 steps; timed: the rest
    # (32 warm-up steps: the other mode plans its GEMMs differently, so the first epochs grow workspaces once more -- a one-off
    # ~80 ms that 8 steps did not always absorb)
    warm_e, warm_c, timed_e, timed_c, left = [], [], [], [], 32
    for plan, cnt in flat:
        k = min(left, len(plan))
        if k:
            warm_e.append(plan[:k]); warm_c.append(cnt[:k])
        if len(plan) > k:
            timed_e.append(plan[k:]); timed_c.append(cnt[k:])
        left -= k
    run(warm_e, warm_c)
    torch.cuda.synchronize()
    # epoch by epoch (a synchronisation per 12 steps): the median epoch is what is reported -- one run in four showed a single
    # ~80 ms host stall somewhere in 200 un-synchronised steps of this secondary phase (never reproduced step by step)
    rates, nodes, steps, out3 = [], 0, 0, None
    t0 = time.perf_counter()
    for pe, pc in zip(timed_e, timed_c):
        torch.cuda.synchronize()
        td = time.perf_counter()
        nn, out3 = run([pe], [pc])
        torch.cuda.synchronize()
        dt = time.perf_counter() - td
        rates.append((dt / len(pe), nn / len(pe)))
        nodes += nn
        steps += len(pe)
    el = time.perf_counter() - t0
    rates.sort()
    ms_med, nodes_med = rates[len(rates) // 2][0] * 1e3, rates[len(rates) // 2][1]      # that epoch's own nodes per step
    out["other_mode"] = {"mode": names[other], "value": nodes_med / (ms_med * 1e-3), "unit": "nodes/s", "steps": steps,
                         "ms_per_step": ms_med, "how": "median over epochs of 12 steps, one synchronisation per epoch",
                         "ms_per_step_all": el / steps * 1e3, "final_loss": float(out3[0])}
    ops.set_gemm_mode(cur)
    return out


def secondary_probe(args, gte, S, dev, pages13):
    """SURVEY 8(d) cfg2, secondary width: the same loop with BBOX features only (F0 = 13; 16 of the reference's 96
    ablation runs).  Same model sizes otherwise; a short run of its own."""
    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd.models import loop
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    resident = G.ResidentPages(to_page_graphs(gte, pages13), dev)
    pipe = loop.BatchPipeline(resident)
    torch.manual_seed(42)
    model = gte.GcnSAGE(13, args.hidden, 9, args.layers, torch.nn.functional.relu, 0).to(dev)
    trainer = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
    sizes = resident.page_sizes()
    warm, e = epoch_steps(sizes, args.pages, 42, 0, 8)
    timed_loop(trainer, pipe, warm, loop)
    steps = 40
    el, nodes, out3 = None, 0, None
    for _ in range(3):                          # (the best of three loops, as the shape probes: one loop is 13 ms)
        epochs, e = epoch_steps(sizes, args.pages, 42, e, steps)
        el_, nodes_, out3 = timed_loop(trainer, pipe, epochs, loop)
        if el is None or nodes_ / el_ > nodes / el:
            el, nodes = el_, nodes_
    dims = [13] + [args.hidden] * (args.layers - 1) + [9]
    flops_node = sum(2.0 * 2 * dims[l] * dims[l + 1] * (3 if l > 0 else 2) for l in range(args.layers))
    out = {"workload": f"cfg2 secondary: F0=13 (BBOX features only), {args.pages} pages per step, hidden={args.hidden}; "
                       f"train loop (a different device-built batch per step)",
           "value": nodes / el, "unit": "nodes/s", "steps": steps, "ms_per_step": el / steps * 1e3,
           "final_loss": float(out3[0]), "mfma_bound_nodes_per_s": MFMA_F32_PEAK_TF * 1e12 / flops_node,
           "frac_of_mfma_bound": nodes / el / (MFMA_F32_PEAK_TF * 1e12 / flops_node)}
    # HIP-graph replay of two resident batches (no batch assembly, no per-kernel launches): the device-time floor
    fixed = [resident.batch(ids) for ids in epochs[0][:2]]
    replays = [trainer.capture(g, g.ndata["label"]) for g in fixed]
    for i in range(8):
        replays[i % 2]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        replays[i % 2]()
    torch.cuda.synchronize()
    el2 = time.perf_counter() - t0
    n2 = sum(fixed[i % 2].num_nodes() for i in range(steps))
    out["replay_nodes_per_s"] = n2 / el2
    out["replay_ms_per_step"] = el2 / steps * 1e3
    trainer.release()
    return out


# The reference's OWN run shapes (run_multiple_train.sh:8-113 of the reference with model_train.py:81-91,157 and
# components/features/utils.py:90-101): --h_layer_dim=1000, or --mode_params=scaled --params_no=100000 -> int(calculate_hidden).
# (363, 139) is the pair the round-3 review named; the grid's own pair for F0 = 363 is (363, 149).
RUN_SHAPES = [(831, 1000), (13, 1000), (13, 218), (363, 149), (363, 139), (831, 96), (781, 100), (313, 157), (63, 206)]
SLICED_FROM = {781: 831, 313: 363, 63: 363}      # page sets of these input widths = the leading columns of a generated set


def _c16(x):
    return -(-int(x) // 16) * 16


def step_bounds(dims, deg, peak_tf, world=1):
    """Whole-step bounds of SURVEY 8(d): per node, layer l (F_l -> F_{l+1}, mean in-degree d), training mode, no recomputation.
    Returns (flops per node, bytes per node, MFMA-bound nodes/s, HBM-bound nodes/s)."""
    n_l = len(dims) - 1
    flops_node = sum(2.0 * 2 * dims[l] * dims[l + 1] * (3 if l > 0 else 2) for l in range(n_l))
    bytes_node = sum(4.0 * (2 * dims[l] + dims[l + 1]) + 8 * deg + 4 +                            # forward
                     4.0 * (dims[l + 1] + 2 * dims[l] + (dims[l] if l > 0 else 0)) + 8 * deg + 4  # backward
                     for l in range(n_l))
    return flops_node, bytes_node, peak_tf * 1e12 / flops_node * world, HBM_PEAK_GBS * 1e9 / bytes_node * world


def forward_gemm_rate(ops, n, fin, fout, dev, aggregate_first=False, reps=10):
    """The forward transform of one hidden layer as the step launches it -- t = h [W_s ; W_n]^T (transform-first: N = 2 fout, K = fin)
    or z = [x | ahn] W^T (aggregate-first: N = fout, K = 2 fin) -- on P3 operands of the step's shapes, `reps` launches inside one
    HIP-event pair on the launch stream.  Returns (ms per launch, fp32-equivalent flops per launch)."""
    g = torch.Generator(device="cpu").manual_seed(3)
    a = ops.p3_from_f32(torch.randn(n, fin, generator=g).to(dev))
    if aggregate_first:
        kp = _c16(fin)
        wb = torch.zeros(fout, 2 * kp)
        wb[:, :fin], wb[:, kp:kp + fin] = torch.randn(fout, fin, generator=g) * 0.05, torch.randn(fout, fin, generator=g) * 0.05
        b = ops.p3_from_f32(wb.to(dev))
        a2 = ops.p3_from_f32(torch.randn(n, fin, generator=g).to(dev))
        out = torch.empty((n, _c16(fout)), dtype=torch.float32, device=dev)
        call = lambda: ops.gemm_p3_nt(a, ops.P3(b.data, fout, 2 * kp), a2=a2, out=out[:, :fout])
        flops = 2.0 * n * 2 * fin * fout
    else:
        ld = _c16(fout)
        b = ops.p3_from_f32((torch.randn(2 * ld, fin, generator=g) * 0.05).to(dev))
        out = torch.empty((n, 2 * ld), dtype=torch.float32, device=dev)
        call = lambda: ops.gemm_p3_nt(a, b, out=out)
        flops = 2.0 * n * fin * 2 * fout
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, flops


def loop_forward_gemms(trainer, pipe, epochs, loop, f0, hid, kinds, n_global=None):
    """The forward transform GEMMs of the two hidden layers INSIDE the train loop: HIP events recorded by the one-call step on its
    launch stream around those launches (gte_step_plan.fwd_events), the steps of ``epochs`` one at a time.  A layer whose LayerNorm
    runs as the GEMM's epilogue (the cached-aggregate input layer) is timed with it.  -> (ms [2], flops [2], steps)"""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    trainer.fwd_events = evs
    ms, fl, steps = [0.0, 0.0], [0.0, 0.0], 0
    try:
        for e, plan_ in enumerate(epochs):
            for k, ids in enumerate(plan_):
                loop.run_steps(trainer, pipe, [ids], n_global=None if n_global is None else [n_global[e][k]])
                torch.cuda.synchronize()
                nn = pipe.nodes(0)
                steps += 1
                for li in range(2):
                    if kinds[li] == 1:
                        continue
                    ms[li] += evs[2 * li].elapsed_time(evs[2 * li + 1])
                    fl[li] += 2.0 * nn * (f0 if li == 0 else hid) * 2 * hid
    finally:
        trainer.fwd_events = None
    return ms, fl, steps


def shapes_probe(args, gte, dev, page_sets, loop):
    """The train loop (the same run_steps as the headline) on the shapes the reference's shipped runs use.  Per shape: nodes/s,
    ms/step, the layer plan the engine chose, and the forward transform GEMMs of the two hidden layers in isolation on the
    step's operand shapes (fp32-equivalent TFLOP/s against the bf16 matrix peak / 6)."""
    from gnn_tableextraction_amd import graph as G, ops
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    out = {"how": f"{args.pages} pages per step, 12 warm-up + 40 timed steps of models/loop.py: run_steps per shape (the best of three such loops); forward GEMMs: "
                  "10 isolated launches per HIP-event pair on operands of the step's shapes",
           "source": "run_multiple_train.sh:8-113 (--h_layer_dim=1000 | --mode_params=scaled --params_no=100000)"}
    peak = MFMA_BF16_PEAK_TF / 6.0
    cache = {}
    for f0, hid in RUN_SHAPES:
        pages = page_sets.get(f0)
        if pages is None and page_sets.get(SLICED_FROM.get(f0)) is not None:
            import dataclasses
            pages = [dataclasses.replace(p, feat=np.ascontiguousarray(p.feat[:, :f0])) for p in page_sets[SLICED_FROM[f0]]]
        if pages is None:
            continue
        key = f0
        if key not in cache:
            cache.clear()
            torch.cuda.empty_cache()
            res = G.ResidentPages(to_page_graphs(gte, pages), dev)
            cache[key] = (res, loop.BatchPipeline(res))
        res, pipe = cache[key]
        torch.manual_seed(42)
        model = gte.GcnSAGE(f0, hid, 9, args.layers, torch.nn.functional.relu, 0).to(dev)
        trainer = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
        sizes = res.page_sizes()
        warm, e = epoch_steps(sizes, args.pages, 42, 0, 12)
        timed_loop(trainer, pipe, warm, loop)
        steps = 40                                              # (round 5: 20 steps x 2 loops read +-8 % from run to run on the 0.3 ms shapes)
        el, nodes, out3 = None, 0, None
        for _ in range(3):                                      # (the best of three loops: a one-off stall -- a buffer set that grows,
            epochs, e = epoch_steps(sizes, args.pages, 42, e, steps)      # a busy host -- is tens of ms against a 15 - 120 ms loop)
            el_, nodes_, out3 = timed_loop(trainer, pipe, epochs, loop)
            if el is None or nodes_ / el_ > nodes / el:
                el, nodes = el_, nodes_
        n_step = int(nodes / steps)
        kinds = trainer._plan_kinds(f0, n_step, res.agg_p3 is not None)
        gen, out_gemm = trainer._plan_mode(kinds, f0) if kinds is not None else (None, None)
        dims = [f0] + [hid] * (args.layers - 1) + [9]
        deg = float(sum(len(p.src) for p in pages)) / max(float(sum(p.num_nodes for p in pages)), 1.0)
        flops_node, bytes_node, mfma_b, hbm_b = step_bounds(dims, deg, peak)
        entry = {"value": nodes / el, "unit": "nodes/s", "ms_per_step": el / steps * 1e3, "nodes_per_step": n_step,
                 "final_loss": float(out3[0]),
                 "plan": None if kinds is None else {"layer_kinds": kinds, "kinds": "0 planes transform-first, 1 one-pass short input, "
                                                     "2 planes aggregate-first, 3 cached aggregate (two resident images)", "padded_rows": bool(gen),
                                                     "output_layer": "planes GEMMs" if out_gemm else "narrow kernels"},
                 "step_tflops_fp32_eq": nodes / el * flops_node / 1e12,
                 "mfma_bound_nodes_per_s": mfma_b, "hbm_bound_nodes_per_s": hbm_b, "bytes_per_node": bytes_node,
                 "bound": "mfma" if mfma_b < hbm_b else "hbm",            # the BINDING bound of SURVEY 8(d): min of the two
                 "frac_of_bound": nodes / el / min(mfma_b, hbm_b),
                 "frac_of_mfma_bound": nodes / el / mfma_b}
        if kinds is not None and trainer._planes_on():
            # the forward transform GEMMs of the two hidden layers INSIDE the loop: HIP events recorded by the one-call step on its
            # launch stream around those launches (gte_step_plan.fwd_events), eight more steps of the same loop, one at a time
            more, e = epoch_steps(sizes, args.pages, 42, e, 8)
            ms, fl, _ = loop_forward_gemms(trainer, pipe, more, loop, f0, hid, kinds)
            tot_ms, tot_fl = ms[0] + ms[1], fl[0] + fl[1]
            entry["forward_gemms"] = {"how": "HIP events on the launch stream around the two hidden layers' forward GEMM launches, 8 steps "
                                             "of the same loop (an event pair costs ~5 us of the interval it brackets)",
                                      "layer0_ms": ms[0] / 8, "layer1_ms": ms[1] / 8, "tflops_fp32_eq": tot_fl / (tot_ms * 1e-3) / 1e12,
                                      "frac_of_peak": tot_fl / (tot_ms * 1e-3) / 1e12 / peak, "peak_tflops": peak,
                                      "layer0_tflops": fl[0] / (ms[0] * 1e-3) / 1e12 if ms[0] > 0 else None,
                                      "layer1_tflops": fl[1] / (ms[1] * 1e-3) / 1e12}
            iso0 = forward_gemm_rate(ops, n_step, f0, hid, dev, aggregate_first=kinds[0] == 2) if kinds[0] != 1 else (0.0, 0.0)
            iso1 = forward_gemm_rate(ops, n_step, hid, hid, dev)
            entry["forward_gemms"]["isolated_random_operands"] = {
                "layer0_ms": iso0[0], "layer1_ms": iso1[0],
                "tflops_fp32_eq": (iso0[1] + iso1[1]) / ((iso0[0] + iso1[0]) * 1e-3) / 1e12,
                "note": "dense N(0,1) operands, 10 launches per event pair: the matrix pipe runs at the power limit, and the loop's "
                        "post-ReLU activations (half zeros) toggle fewer bits"}
        if (f0, hid) == (831, 1000):
            # the same loop with the fp32 MFMA kernels
            prev = ops.set_gemm_mode("f32")
            try:
                torch.manual_seed(42)
                m2 = gte.GcnSAGE(f0, hid, 9, args.layers, torch.nn.functional.relu, 0).to(dev)
                t2 = FusedGcnSageStep(m2, lr=0.01, weight_decay=5e-4)
                w2, e2 = epoch_steps(sizes, args.pages, 42, 0, 4)
                timed_loop(t2, pipe, w2, loop)
                ep2, e2 = epoch_steps(sizes, args.pages, 42, e2, 10)
                el2, nodes2, _ = timed_loop(t2, pipe, ep2, loop)
                entry["f32_mode"] = {"value": nodes2 / el2, "unit": "nodes/s", "ms_per_step": el2 / 10 * 1e3}
                del t2, m2
            finally:
                ops.set_gemm_mode(prev)
        out[f"f{f0}_h{hid}"] = entry
        del trainer, model
    cache.clear()
    torch.cuda.empty_cache()
    return out


def uncached_probe(args, gte, dev, pages, loop):
    """The headline loop WITHOUT the cached input aggregate (GTE_CACHE_AGG=0's path): layer 0 aggregates its input in every step
    (reference models.py:53-54 as written) instead of reading the per-page aggregate made once at load.  Same model, same pages."""
    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    torch.cuda.empty_cache()
    res = G.ResidentPages(to_page_graphs(gte, pages[:600]), dev)
    pipe = loop.BatchPipeline(res)
    torch.manual_seed(42)
    model = gte.GcnSAGE(args.in_feats, args.hidden, 9, args.layers, torch.nn.functional.relu, 0).to(dev)
    trainer = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
    trainer.cache_input_agg = False
    sizes = res.page_sizes()
    warm, e = epoch_steps(sizes, args.pages, 42, 0, 12)
    timed_loop(trainer, pipe, warm, loop)
    steps, el, nodes = 40, None, 0
    for _ in range(3):
        epochs, e = epoch_steps(sizes, args.pages, 42, e, steps)
        el_, nodes_, _ = timed_loop(trainer, pipe, epochs, loop)
        if el is None or nodes_ / el_ > nodes / el:
            el, nodes = el_, nodes_
    kinds = trainer._plan_kinds(args.in_feats, int(nodes / steps), res.agg_p3 is not None)
    out = {"workload": "the headline loop with the input's mean aggregate computed in every step (no cached aggregate image)",
           "how": "12 warm-up + 40 timed steps of run_steps, the best of three such loops, 600 resident pages",
           "value": nodes / el, "unit": "nodes/s", "ms_per_step": el / steps * 1e3, "layer_kinds": kinds}
    del trainer, model, pipe, res
    torch.cuda.empty_cache()
    return out


def size_sweep_probe(args, trainer, pipe, sizes, loop, first_epoch):
    """nodes/s of the headline configuration at other batch sizes (pages per step): real PubLayNet pages give 2 x 10^4 ... 8 x 10^4
    nodes per 100-page step; the time of a step should follow its node count, not the number of tile rounds."""
    out = {"how": "8 warm-up + 32 timed steps of the same loop per point (the best of three such loops), same resident pages", "points": []}
    e = first_epoch
    for pages_per_step in (50, 100, 135, 150, 200, 330):
        if pages_per_step > len(sizes):
            continue
        warm, e = epoch_steps(sizes, pages_per_step, 42, e, 8)
        timed_loop(trainer, pipe, warm, loop)
        steps = 32
        el, nodes = None, 0
        for _ in range(3):                  # (a buffer set that grows inside a timed loop -- a batch larger than every warm-up
            ep, e = epoch_steps(sizes, pages_per_step, 42, e, steps)       # batch -- costs tens of ms once: the best of three loops)
            el_, nodes_, _ = timed_loop(trainer, pipe, ep, loop)
            if el is None or nodes_ / el_ > nodes / el:
                el, nodes = el_, nodes_
        out["points"].append({"pages_per_step": pages_per_step, "nodes_per_step": int(nodes / steps), "ms_per_step": el / steps * 1e3,
                              "value": nodes / el, "unit": "nodes/s"})
    best = max(p["value"] for p in out["points"])
    for p in out["points"]:
        p["frac_of_best"] = p["value"] / best
    out["min_frac_of_best"] = min(p["frac_of_best"] for p in out["points"])
    hl = [p["frac_of_best"] for p in out["points"] if p["pages_per_step"] == args.pages]
    out["headline_frac_of_best"] = hl[0] if hl else None          # the reference's batch size (parsers/graphs.py:74)
    return out


def residency_probe(args, gte, dev, pages, loop):
    """A training set that does NOT fit its HBM budget (models/residency.py): the pages in pinned host memory, two window slots on
    the device, the next window uploaded on a copy stream while the current one trains, `passes` passes per window visit.  Same
    model, same step stream on the all-resident set for the reference rate."""
    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd.models import residency as R
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    graphs = to_page_graphs(gte, pages)
    n_steps = 240

    def fresh():
        torch.manual_seed(42)
        m = gte.GcnSAGE(args.in_feats, args.hidden, 9, args.layers, torch.nn.functional.relu, 0).to(dev)
        return FusedGcnSageStep(m, lr=0.01, weight_decay=5e-4)
    tr = fresh()
    want_p3 = tr.wants_p3_features(args.in_feats)
    want_agg = bool(want_p3 and tr.wants_agg_image(args.in_feats))       # (the cached aggregate of the input travels with every window)
    t0 = time.perf_counter()
    host = R.HostPages(graphs, dev)
    build_s = time.perf_counter() - t0
    per_node = R.WindowedPages.bytes_per_node(host.page_nodes, host.page_edges, args.in_feats, want_p3, want_agg)
    set_bytes = float(host.page_nodes.sum()) * per_node
    budget = set_bytes / 2.5                                     # two slots of ~1/6 of the set each (+ the staging rows)
    wp = R.WindowedPages(host, budget, want_p3, want_agg)
    # raw pinned-host -> device rate of this box (one window's feature rows)
    p0, p1 = wp.ranges[0]
    n0, n1 = int(host.node_off[p0]), int(host.node_off[p1])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst = torch.empty((n1 - n0, args.in_feats), dtype=torch.float32, device=dev)
    dst.copy_(host.feat[n0:n1], non_blocking=True)
    e0.record()
    dst.copy_(host.feat[n0:n1], non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    h2d = (n1 - n0) * args.in_feats * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del dst
    out = {"workload": f"{len(pages)} pages ({set_bytes / 1e9:.2f} GB in resident form) under a budget of {budget / 1e9:.2f} GB: "
                       f"{len(wp.ranges)} windows, {args.pages} pages per step",
           "host_build_s": build_s, "pinned_h2d_GB_per_s": h2d, "device_bytes": wp.device_bytes, "windowed": {}}
    for passes in (1, 4, 8, 16):
        tr = fresh()
        stream = R.WindowStream(wp.ranges, args.pages, passes, 42)
        wp.prefetch(stream.peek_window())
        pipe = loop.BatchPipeline(wp.acquire(stream.peek_window()))
        pipe._bound_pages = (host.page_nodes, np.diff(host.sets["in"]["edge_off"]), np.diff(host.sets["out"]["edge_off"]))
        R.run_windowed(tr, pipe, wp, stream, 24)                     # warm-up
        torch.cuda.synchronize()
        best = None
        for _ in range(2):              # (the better of two runs of n_steps: the host link is shared with the node's other jobs)
            up0 = wp.uploaded_bytes
            nodes = [0]
            t0 = time.perf_counter()
            ht = {}
            R.run_windowed(tr, pipe, wp, stream, n_steps, on_step=lambda s, g, o: nodes.__setitem__(0, nodes[0] + g.num_nodes()),
                           host_times=ht)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            entry = {"value": nodes[0] / el, "unit": "nodes/s", "ms_per_step": el / n_steps * 1e3, "steps": n_steps,
                     "upload_GB_per_s": (wp.uploaded_bytes - up0) / el / 1e9,
                     "host_ms": {"queueing_total": t_host * 1e3, **{k: (v * 1e3 if k != "chunks" else v) for k, v in ht.items()}}}
            if best is None or entry["value"] > best["value"]:
                best = entry
        out["windowed"][f"passes_{passes}"] = best
        del tr, pipe
    # the same kind of stream on the all-resident set
    tr2 = fresh()
    res = G.ResidentPages(graphs, dev)
    if want_p3:
        res.enable_p3(agg=want_agg)
    pipe2 = loop.BatchPipeline(res)
    stream2 = R.WindowStream(wp.ranges, args.pages, 4, 42)

    def run_resident(k):
        n = 0
        for w, steps in stream2.take(k):
            p0 = wp.ranges[w][0]
            loop.run_steps(tr2, pipe2, [ids + p0 for ids in steps])
            n += sum(pipe2.nodes(i) for i in range(len(steps)))
        return n
    run_resident(24)
    torch.cuda.synchronize()
    best2 = None
    for _ in range(2):                                           # (the better of two runs, like the windowed settings)
        t0 = time.perf_counter()
        n2 = run_resident(n_steps)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        if best2 is None or n2 / el2 > best2[0] / best2[1]:
            best2 = (n2, el2)
    n2, el2 = best2
    out["all_resident"] = {"value": n2 / el2, "unit": "nodes/s", "ms_per_step": el2 / n_steps * 1e3}
    for k, v in out["windowed"].items():
        v["over_all_resident"] = v["value"] / out["all_resident"]["value"]
    out["note"] = ("a step consumes its batch's feature rows at ~147 GB/s; the host link delivers pinned_h2d_GB_per_s: a window has to "
                   "be trained on for several passes per upload (GTE_WINDOW_PASSES) before the upload hides behind it")
    return out


def cfg3_probe(args, gte, dev):
    """BASELINE configs[2]: "PubTables-1M table-structure graphs, 4-head GAT bf16" -- no reference counterpart (the
    reference has no GAT and builds no table graphs), PARITY UNPINNED (oracle = oracle/gat_cpu.py, the build's own restatement).
    Synthetic table-cell grids (R ~ U{3..40} rows x C ~ U{2..12} columns, edges to row / column neighbours, both directions),
    GAT 3 layers x 4 heads x 64, bf16 projection (v_mfma_f32_32x32x16_bf16, fp32 accumulate) + bf16 gathers; a step = forward +
    CE + backward + Adam."""
    from gnn_tableextraction_amd import graph as G, ops, _lib
    rng = np.random.default_rng(3)
    srcs, dsts, off = [], [], 0
    while off < 200_000:
        R, C = int(rng.integers(3, 41)), int(rng.integers(2, 13))
        idx = np.arange(R * C).reshape(R, C)
        pairs = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()]), np.stack([idx[:-1].ravel(), idx[1:].ravel()])], 1)
        pairs = np.concatenate([pairs, pairs[::-1]], 1)
        srcs.append(pairs[0] + off); dsts.append(pairs[1] + off); off += R * C
    src, dst, n = np.concatenate(srcs), np.concatenate(dsts), off
    f_in, heads, hid, ncls = 32, 4, 64, 5
    torch.manual_seed(0)
    g = G.PageGraph(src, dst, n, device=dev)
    x = torch.randn(n, f_in, device=dev)
    y = torch.from_numpy(rng.integers(0, ncls, n)).to(dev)
    out = {"workload": f"cfg3: {n} table-cell nodes / {len(src)} edges (synthetic grids), GAT 3 layers x {heads} heads x {hid}, "
                       f"F0={f_in}, {ncls} classes; parity unpinned (no reference GAT)"}
    for tag, cd, gd in (("bf16", torch.bfloat16, torch.bfloat16), ("f32", torch.float32, torch.float32)):
        torch.manual_seed(1)
        model = gte.GAT(f_in, hid, ncls, n_layers=3, heads=heads, gather_dtype=gd, compute_dtype=cd).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=0.01)

        def step():
            loss, _ = ops.cross_entropy(model(g, x), y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            return loss
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            loss = step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 20
        out[tag] = {"ms_per_step": el * 1e3, "nodes_per_s": n / el, "final_loss": float(loss)}
    # the two kernels the precision choice changes, alone: hidden-layer projection (K = N = 256) and its aggregation
    lib, P, st = _lib.load(), _lib.ptr, _lib.current_stream()
    hd = heads * hid
    h = torch.randn(n, hd, device=dev)
    w = torch.randn(hd, hd, device=dev) * 0.05
    from gnn_tableextraction_amd.components.graphs.gat import _bf16_copy
    hb, wb = _bf16_copy(h), _bf16_copy(w)
    z = torch.empty(n, hd, device=dev)

    def ev_time(fn, reps=20):
        for _ in range(5):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps
    ms_b = ev_time(lambda: lib.gte_gemm_bf16_nt(P(hb), hd, P(wb), hd, P(z), hd, n, hd, hd, st))
    ms_f = ev_time(lambda: ops.gemm(h, w, trans_b=True, out=z))
    flops = 2.0 * n * hd * hd
    bytes_b = n * hd * (2 + 4) + hd * hd * 2
    out["projection_256x256"] = {"kernel": "gemm_bf16_nt_kernel (v_mfma_f32_32x32x16_bf16, fp32 accumulate)",
                                 "bf16_ms": ms_b, "bf16_TFLOPs": flops / ms_b / 1e9, "f32_mfma_ms": ms_f, "f32_TFLOPs": flops / ms_f / 1e9,
                                 "roofline": {"bound": "hbm", "achieved": bytes_b / ms_b / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": bytes_b / ms_b / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes": bytes_b,
                                              "note": "2 K N flop per (2 K + 4 N) bytes = 85 flop/B, bf16 ridge ~310: HBM-bound"}}
    csr = g.in_csr()
    a_l, a_r = torch.randn(hd, device=dev) * 0.1, torch.randn(hd, device=dev) * 0.1
    el_, er_ = torch.empty(n, heads, device=dev), torch.empty(n, heads, device=dev)
    zb = torch.empty(n, hd, dtype=torch.bfloat16, device=dev)
    lib.gte_gat_scores(P(z), hd, P(a_l), P(a_r), P(el_), P(er_), P(zb), hd, n, heads, hid, st)
    o32, ob = torch.empty(n, hd, device=dev), torch.empty(n, hd, dtype=torch.bfloat16, device=dev)
    smax, ssum = torch.empty_like(el_), torch.empty_like(el_)
    ms_a = ev_time(lambda: lib.gte_gat_aggregate_fwd_ex(P(csr.indptr), P(csr.indices), P(zb), hd, 1, P(el_), P(er_), None, P(o32), hd,
                                                        P(smax), P(ssum), n, heads, hid, 1, P(ob), hd, None, 0, None, st))
    e_cnt = csr.indices.numel()
    bytes_a = n * hd * (2 + 4 + 2) + e_cnt * (4 + heads * 4) + n * (4 + heads * 16)
    out["aggregation_256"] = {"kernel": "gat_rows_fwd_kernel<bf16, 4> (head per DPP row, online edge softmax per 16-edge chunk, ELU + bf16 copy in the epilogue)",
                              "ms": ms_a, "roofline": {"bound": "hbm", "achieved": bytes_a / ms_a / 1e6, "peak": HBM_PEAK_GBS,
                                                       "unit": "GB/s", "frac": bytes_a / ms_a / 1e6 / HBM_PEAK_GBS,
                                                       "algorithmic_bytes": bytes_a}}
    return out


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
        for _ in range(12):            # (enough launches for the sustained clocks: the probe is the process's first GPU work)
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
    # (ii) of SURVEY 8(d) cfg4: the backward pass of the same aggregation = a SUM over the OUT-edge CSR with the weights
    # w / in_degree(dst) (d(norm * A_w h) / dh), same kernel family, its own tile plan
    dst_t, src_t = torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev)
    w_bwd = torch.from_numpy(w).to(dev) * ops.inv_degree(indptr)[dst_t.long()]
    r_indptr, r_indices, _, r_w = ops.coo_to_csr(src_t, dst_t, n, w_bwd)
    r_plan = ops.build_tile_plan(r_indptr, r_indices, n)
    # (building the reverse CSR and its tile plan left the GPU idle for a few hundred ms: the first launches after that run at
    # lower clocks -- 1040 us against 850 us once warm, profiles/debug/gather_timing_methods.py; 20 warm-up launches ~ 18 ms)
    bwd_evs = []
    for i in range(30):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        ops.spmm_csr(r_indptr, r_indices, r_w, x, n, mean=False, out=out, tiles=r_plan, force_tiled=True)
        e_.record()
        if i >= 20:
            bwd_evs.append((s_, e_))
    torch.cuda.synchronize()
    bwd_ms = float(np.mean([s_.elapsed_time(e_) for s_, e_ in bwd_evs]))
    del r_plan, r_indptr, r_indices, r_w, w_bwd
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
            "bwd_ms_per_pass": bwd_ms, "bwd_GBs": alg_bytes / (bwd_ms * 1e-3) / 1e9, "bwd_frac": alg_bytes / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "device_copy_same_matrix_ms": copy_ms, "device_copy_GBs": 2.0 * n * f * 4 / (copy_ms * 1e-3) / 1e9,
            "full_layer_512_fwd_bwd_ms": layer_ms,
            "full_layer_nodes_per_s": (n / (layer_ms * 1e-3)) if layer_ms else None,
            "traffic": pmc_traffic()[0].get("gather_cfg4_tiled_bytes_per_launch") if n == 1_000_000 else None,
            "traffic_source": pmc_traffic()[1]}

RECORD_MAX_CHARS = 3000      # the driver keeps ~8 000 characters of stdout; round 4's 22 kB line was cut and could not be parsed


def _r(x, digits=4):
    """floats to `digits` significant digits (the record is a summary: the full precision is in bench_extras.json)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def compact_record(full):
    """The ONE stdout line: the driver's keys, `roofline`, `step_roofline`, `cpu_baseline`, and one-number summaries of the
    secondary measurements.  No prose beyond `config.workload`; everything else lives in bench_extras.json."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    rec = {k: full[k] for k in keep}
    cfg = full["config"]
    rec["config"] = {"workload": (f"cfg2: {cfg['pages_per_gpu_per_step']} synthetic PubLayNet-style page graphs per GPU per step "
                                  f"(~{cfg['nodes_per_step_per_gpu']} nodes), GcnSAGE {cfg['layers']} layers F0={cfg['in_feats']} "
                                  f"hidden={cfg['hidden']} classes=9 fp32, CE + Adam; train loop, a different device-built batch "
                                  f"every step" + ("; layer-0 input aggregate cached per resident page (made once at load, not in the "
                                                   "timed step; `uncached` = the same loop aggregating the input every step)"
                                                   if cfg.get("input_aggregate_cached") else "")),
                     "pages_per_gpu_per_step": cfg["pages_per_gpu_per_step"], "nodes_per_step_per_gpu": cfg["nodes_per_step_per_gpu"],
                     "resident_pages_per_gpu": cfg["resident_pages_per_gpu"], "parallelism": cfg["parallelism"]}
    ro = full["roofline"]
    rec["roofline"] = {"bound": ro["bound"], "achieved": ro["achieved"], "peak": ro["peak"], "unit": ro["unit"], "frac": ro["frac"],
                       "traffic": ro["traffic"], "kernel": ro["kernel_short"], "avg_launch_ms": ro["avg_launch_ms"],
                       "launches": ro["launches"], "algorithmic_flops_per_launch": ro["algorithmic_flops_per_launch"]}
    if ro.get("layer1"):
        rec["roofline"]["layer1"] = ro["layer1"]
    sr = full["step_roofline"]
    rec["step_roofline"] = {k: sr[k] for k in ("bound", "frac", "flops_per_node", "bytes_per_node", "mfma_bound_nodes_per_s",
                                                "hbm_bound_nodes_per_s", "traffic_bytes_per_node", "traffic_GBs") if k in sr}
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        rec["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": cb["sample_short"], "ms_per_step": cb["ms_per_step"]}
        rec["gpu_over_cpu"] = full["gpu_over_cpu"]
    rec["final_loss"] = full["final_loss"]
    if isinstance(full.get("uncached"), dict) and "value" in full["uncached"]:
        rec["uncached"] = full["uncached"]["value"]
    if "long_run" in full:
        rec["long_run"] = {k: full["long_run"][k] for k in ("steps", "seconds", "value")}
    if "gather" in full:
        g = full["gather"]
        rec["gather"] = {"frac": g["frac"], "bwd_frac": g["bwd_frac"], "GBs": g["achieved"], "traffic": g["traffic"],
                         "algorithmic_bytes": g["algorithmic_bytes"]}
    if "shapes" in full:
        rec["shapes"] = {k: [v["value"], v["frac_of_bound"]] for k, v in full["shapes"].items() if isinstance(v, dict) and "value" in v}
        rec["shapes_fmt"] = "f<F0>_h<H>: [nodes/s, frac of min(HBM, MFMA) step bound]"
    if "size_sweep" in full:
        rec["size_sweep"] = {"min_frac_of_best": full["size_sweep"]["min_frac_of_best"],
                             "headline_frac_of_best": full["size_sweep"].get("headline_frac_of_best")}
    if "residency" in full:
        w = full["residency"]["windowed"]
        rec["residency"] = {"over_all_resident": {k: v["over_all_resident"] for k, v in w.items()}}
    for k in ("replay", "secondary"):
        if k in full and isinstance(full[k], dict) and "value" in full[k]:
            rec[k] = full[k]["value"]
    if isinstance(full.get("val_graph"), dict) and "nodes_per_s" in full["val_graph"]:
        rec["val_graph"] = full["val_graph"]["nodes_per_s"]
    if "gemm_modes" in full:
        rec["f32_mfma_mode"] = full["gemm_modes"]["other_mode"]["value"]
    if isinstance(full.get("dist_1rank"), dict) and "extra_us" in full["dist_1rank"]:
        rec["dist_1rank_extra_us"] = full["dist_1rank"]["extra_us"]
    rec["extras"] = "bench_extras.json"
    rec = {k: (_r(v, 7) if k in keep else _r(v)) for k, v in rec.items()}
    s = json.dumps(rec, separators=(",", ":"))
    # never let the record outgrow the driver's window: drop the secondary summaries first
    for k in ("shapes_fmt", "val_graph", "secondary", "replay", "f32_mfma_mode", "residency", "size_sweep", "shapes", "gather", "long_run"):
        if len(s) <= RECORD_MAX_CHARS:
            break
        rec.pop(k, None)
        s = json.dumps(rec, separators=(",", ":"))
    return s


def emit(full):
    """Full measurements -> bench_extras.json (next to this file; also under gpurun_out/ when that exists) and stderr; the compact
    record -> the LAST stdout line."""
    blob = json.dumps(full, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_extras.json"), "w") as f:
                    f.write(blob)
            except OSError as e:                                   # a read-only tree must not cost the record
                print(f"bench.py: could not write bench_extras.json in {d}: {e}", file=sys.stderr)
    print(json.dumps(full), file=sys.stderr, flush=True)
    sys.stdout.flush()
    rec = compact_record(full) + "\n"
    if _RECORD_FD is not None:
        os.write(_RECORD_FD, rec.encode())
    else:
        sys.stdout.write(rec)
        sys.stdout.flush()


_RECORD_FD = None


def main():
    args = parse()
    maybe_spawn(args)                       # --gpus N outside torchrun: N fresh ranks, before any GPU call (does not return)
    # The record must be the LAST line on stdout.  Libraries write there too -- RCCL prints "Librccl path : ..." from C code when the
    # process exits, after everything Python printed -- so file descriptor 1 of every rank is pointed at stderr for the rest of the
    # process and the record alone goes to a private duplicate of the real stdout (emit).
    global _RECORD_FD
    sys.stdout.flush()
    _RECORD_FD = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GTE_BENCH_FORCE_DIST=1 (test hook): run the data-parallel code path -- RCCL process group, flat-gradient all-reduce
    # -- even with one rank, so a 1-GPU box exercises RCCL.
    distributed = world > 1 or os.environ.get("GTE_BENCH_FORCE_DIST", "0") == "1"
    # Test hook for 1-GPU boxes (GTE_BENCH_SHARE_GPU=1): every rank uses cuda:0 and the ranks talk over gloo (RCCL refuses two
    # ranks on one device) -- exercises the N > 1 code path of this file, not a measurement.
    share_gpu = os.environ.get("GTE_BENCH_SHARE_GPU", "0") == "1"

    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd.data import synthetic as S

    # ---- synthetic pages, on host cores, BEFORE the GPU is touched (the pool forks) ---------------------------------
    t_gen = time.perf_counter()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    workers = max(1, usable_cores() // max(local_world, 1))
    pages = make_pages_parallel(args.resident_pages, args.in_feats, rank * args.resident_pages, workers)
    extras = rank == 0 and world == 1 and not distributed
    pages13 = None
    if extras and not args.no_secondary and args.in_feats != 13:
        pages13 = make_pages_parallel(min(args.resident_pages, 600), 13, 0, workers)
    pages_res = None
    if extras and not args.no_residency:
        # the host-resident probe: a set big enough that a window holds several steps' pages (image < 4 GB: the all-resident
        # reference keeps reading it through the row map)
        pages_res = make_pages_parallel(1800, args.in_feats, 20_000_000, workers)
    pages363 = None
    if extras and not args.no_shapes and args.in_feats == 831:
        pages363 = make_pages_parallel(300, 363, 0, workers)
    val_pages = None
    if rank == 0 and world == 1 and args.val_graph > 0:
        val_pages = make_pages_parallel(args.val_graph, args.in_feats, 10_000_000, workers)
    gen_s = time.perf_counter() - t_gen

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
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
        if dist.get_world_size() != max(args.gpus, 1):
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
        world = dist.get_world_size()

    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd import ops
    from gnn_tableextraction_amd.models import loop
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep

    if args.gemm_mode is not None:
        ops.set_gemm_mode(args.gemm_mode)
    split_mode = ops.get_gemm_mode() == ops.GEMM_SPLIT_BF16
    resident = G.ResidentPages(to_page_graphs(gte, pages), dev)        # this rank's pages, resident in HBM
    pipe = loop.BatchPipeline(resident)
    sizes = resident.page_sizes()
    torch.manual_seed(42)
    model = gte.GcnSAGE(args.in_feats, args.hidden, 9, args.layers, torch.nn.functional.relu, 0)
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    trainer = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4, distributed=distributed)

    seed = 42 + rank                                   # every rank shuffles ITS pages
    warm, ep = epoch_steps(sizes, args.pages, seed, 0, args.warmup)
    timed, ep = epoch_steps(sizes, args.pages, seed, ep, args.steps)
    prof, ep = epoch_steps(sizes, args.pages, seed, ep, 8)
    alt, ep = epoch_steps(sizes, args.pages, seed, ep, 32 + 200)      # the other GEMM mode: 32 warm-up + 200 timed steps
    n_long = 0

    def node_counts(epochs):
        return [[int(sum(sizes[i] for i in ids)) for ids in plan] for plan in epochs]

    def global_counts(epochs):
        """per-step node counts over all ranks (every rank plans its own pages: one small all-reduce, outside the timing)"""
        local = node_counts(epochs)
        if not distributed:
            return local
        flat = torch.tensor([c for plan in local for c in plan], dtype=torch.int64, device=dev)
        dist.all_reduce(flat)
        flat = flat.cpu().tolist()
        out, k = [], 0
        for plan in local:
            out.append(flat[k:k + len(plan)])
            k += len(plan)
        return out

    step_events = [] if os.environ.get("GTE_BENCH_STEP_TIMES") else None      # debug: per-step device times on stderr

    def _mark(s, g, o):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        step_events.append((ev, g.num_nodes()))

    epoch_diag = [] if os.environ.get("GTE_BENCH_LONG_DIAG") else None        # debug: (host time, device event) per run_steps call

    def run(epochs, counts):
        nodes, out3 = 0, None
        for plan, cnt in zip(epochs, counts):
            out3 = loop.run_steps(trainer, pipe, plan, n_global=cnt if distributed else None,
                                  on_step=_mark if step_events is not None else None)
            nodes += sum(pipe.nodes(i) for i in range(len(plan)))
            if epoch_diag is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                epoch_diag.append((time.perf_counter(), ev, len(plan)))
        return nodes, out3

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    warm_cnt, timed_cnt, prof_cnt = global_counts(warm), global_counts(timed), global_counts(prof)
    # The set-up above leaves millions of long-lived Python objects behind (page arrays, plans); a full collection walks all of
    # them (tens of ms) whenever the young generations overflow, i.e. in the middle of a timed loop whose host side must
    # sustain a launch every ~30 us.  Collect once and move the survivors out of the collector's sight.
    import gc
    if os.environ.get("GTE_BENCH_GC_FREEZE", "1") == "1":
        gc.collect()
        gc.freeze()
    # The other workloads of the line (cfg4 aggregation, validation graph) run FIRST: they are independent measurements, and
    # the GPU reaches its sustained clocks under them -- a training run is at those clocks; a 20-step region measured
    # seconds after an idle GPU woke up is not (measured: the same 20 steps take 7 % longer right after start-up).
    pre = {}
    if rank == 0 and world == 1 and not args.no_gather_probe:
        pre["gather"] = gather_probe(args, gte, S, dev)
    if val_pages is not None:
        pre["val_graph"] = val_graph_probe(args, gte, S, model, dev, val_pages, trainer)
        val_pages = None
    torch.cuda.empty_cache()
    # (all host-side planning for the warm-up and the timed steps is done above: nothing slow sits between the probes and them)
    # ---- secondary long run of the same loop: >= long_run_seconds of device time ------------------------------------
    long_run = None
    if args.long_run_seconds > 0:
        # sized from the step's algorithmic work at the fp32 matrix peak (a lower bound of its time): >= long_run_seconds
        dims_ = [args.in_feats] + [args.hidden] * (args.layers - 1) + [9]
        flops_node_ = sum(2.0 * 2 * dims_[l] * dims_[l + 1] * (3 if l > 0 else 2) for l in range(args.layers))
        t_floor = float(np.mean(sizes)) * args.pages * flops_node_ / (MFMA_F32_PEAK_TF * 1e12)
        n_long = max(args.steps, int(args.long_run_seconds / max(t_floor, 1e-6)) + 1)
        if distributed:
            # every rank generates ITS pages, so the mean page size -- and with it this step count -- differs by rank; the
            # ranks must run the same number of steps (one all-reduce per step): take the largest
            nl = torch.tensor([n_long], dtype=torch.int64, device=dev)
            dist.all_reduce(nl, op=dist.ReduceOp.MAX)
            n_long = int(nl.item())
        # (its own few warm-up steps: this is the first training activity of the process -- buffer sets, the plan, the first launch
        # of every kernel; measured 0.28 s, which read as 37 M instead of 44 M nodes/s over a 1.4 s run)
        pre_ep, ep = epoch_steps(sizes, args.pages, seed, ep, max(args.warmup, 4))
        run(pre_ep, global_counts(pre_ep))
        if epoch_diag is not None:
            epoch_diag.clear()
        long_ep, ep = epoch_steps(sizes, args.pages, seed, ep, n_long)
        long_cnt = global_counts(long_ep)
        barrier()
        t1 = time.perf_counter()
        nodes_long, _ = run(long_ep, long_cnt)
        barrier()
        el_long = time.perf_counter() - t1
        st = torch.tensor([el_long, float(nodes_long)], dtype=torch.float64, device=dev)
        if distributed:
            mx = st.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(st, op=dist.ReduceOp.SUM)
            el_long, nodes_long = float(mx[0]), float(st[1])
        long_run = {"steps": n_long, "seconds": el_long, "ms_per_step": el_long / n_long * 1e3,
                    "value": nodes_long / el_long, "unit": "nodes/s"}
        if epoch_diag:
            hs = np.diff(np.array([h for h, _, _ in epoch_diag])) * 1e3
            ds = np.array([epoch_diag[i][1].elapsed_time(epoch_diag[i + 1][1]) for i in range(len(epoch_diag) - 1)])
            k = epoch_diag[0][2]
            print(f"long run, per run_steps call of {k} steps: device ms median {np.median(ds):.2f} p90 {np.percentile(ds, 90):.2f} max "
                  f"{ds.max():.2f} (calls above 1.3 x median: {int((ds > 1.3 * np.median(ds)).sum())} of {len(ds)}); host ms median "
                  f"{np.median(hs):.2f} p90 {np.percentile(hs, 90):.2f} max {hs.max():.2f}; device ms by tenth of the run: "
                  + " ".join(f"{c.mean():.2f}" for c in np.array_split(ds, 10)), file=sys.stderr)
            epoch_diag.clear()

    run(warm, warm_cnt)
    barrier()
    t0 = time.perf_counter()
    nodes_local, out3 = run(timed, timed_cnt)
    barrier()
    elapsed = time.perf_counter() - t0
    final_loss = float(out3[0])
    if step_events is not None:
        evs = step_events[-(args.steps + 1):]
        print("per-step us / nodes:", " ".join(f"{evs[i][0].elapsed_time(evs[i + 1][0]) * 1e3:.0f}/{evs[i + 1][1]}"
                                               for i in range(len(evs) - 1)), file=sys.stderr)
        step_events = None

    # ---- per-kernel HIP-event timing: the SAME loop with event pairs around the tagged launches, right after -------------
    kt = {}
    if not args.no_kernel_timers:
        run(prof[:1], prof_cnt[:1])              # untimed: settle
        ops.enable_kernel_timers(True)
        run(prof, prof_cnt)
        kt = ops.kernel_timer_report()
        ops.enable_kernel_timers(False)

    stat = torch.tensor([elapsed, float(nodes_local)], dtype=torch.float64, device=dev)
    if distributed:
        tmax = stat.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)
        elapsed, nodes_total = float(tmax[0]), float(stat[1])
    else:
        nodes_total = float(nodes_local)

    # the forward transform GEMMs of the loop the timed region ran (one-call step: events on its launch stream around them)
    kinds_hl = trainer._plan_kinds(args.in_feats, int(np.mean(sizes)) * args.pages, resident.agg_p3 is not None) if args.layers == 3 else None
    fwd_ev = None
    if kinds_hl is not None and trainer._planes_on():
        more, ep = epoch_steps(sizes, args.pages, seed, ep, 8)
        ms2, fl2, n2 = loop_forward_gemms(trainer, pipe, more, loop, args.in_feats, args.hidden, kinds_hl,
                                          n_global=global_counts(more) if distributed else None)
        # the DOMINANT kernel is ONE kernel: the layer-0 forward GEMM (layer 1's goes beside it as `layer1`, never averaged in);
        # a short-input layer 0 (kind 1) is no GEMM: layer 1's forward is the dominant GEMM then
        dom = 0 if kinds_hl[0] != 1 else 1
        fwd_ev = (n2, ms2[dom], fl2[dom], ms2, fl2, dom)
    if rank == 0:
        n_launch, ms, flops = kt.get("gemm_nt", (0, 0.0, 0.0))
        if fwd_ev is not None and fwd_ev[1] > 0:
            n_launch, ms, flops = fwd_ev[0], fwd_ev[1], fwd_ev[2]
        tf = (flops / (ms * 1e-3) / 1e12) if ms > 0 else 0.0
        # split mode: fp32-equivalent flops (2 M N K) against the bf16 matrix peak / 6 (six bf16 MFMA products per fp32 product)
        gemm_peak = MFMA_BF16_PEAK_TF / 6.0 if split_mode else MFMA_F32_PEAK_TF
        roofline = {"bound": "mfma",
                    "kernel": (("gemm_p3_nt_lw_kernel / gemm_p3_nt_ring_kernel (layer transforms, forward; operands as P3 images -- 3 exact "
                                "bf16 planes written by their producers --, LDS-DMA operand loads, 6 bf16 MFMA products, fp32 accumulate; "
                                "peak = bf16 dense / 6)") if trainer._planes_on() else
                               ("gemm_split_kernel<NT> (layer transforms, forward; fp32 operands as 3 exact bf16 pieces, 6 bf16 "
                                "MFMA products, fp32 accumulate; peak = bf16 dense / 6)")) if split_mode
                              else "gemm_f32_mfma_kernel<NT> (layer transforms, forward)",
                    "kernel_short": ((((f"layer-{fwd_ev[5]} forward GEMM alone" + (" ([x|cached ahn]W^T, K=2*F0, LN+ReLU epilogue)" if kinds_hl[0] == 3 else "")
                                        + ", gemm_p3_nt (bf16x3 planes, peak=bf16/6)") if fwd_ev is not None else "gemm_p3_nt fwd (bf16x3 planes, peak=bf16/6)")
                                      if trainer._planes_on() else "gemm_split<NT> fwd (peak=bf16/6)")
                                     if split_mode else "gemm_f32_mfma<NT> fwd"),
                    "achieved": tf, "peak": gemm_peak, "unit": "TFLOP/s", "frac": tf / gemm_peak,
                    "launches": n_launch, "avg_launch_ms": ms / max(n_launch, 1),
                    "algorithmic_flops_per_launch": flops / max(n_launch, 1),
                    "how": ("HIP events recorded by the one-call step on its launch stream around the forward transform GEMM of each "
                            "hidden layer, 8 steps of the same loop right after the timed region (layer kinds "
                            f"{kinds_hl}: 3 = input layer on [x | cached mean aggregate], K = 2 F0, LayerNorm + ReLU in the GEMM's epilogue)")
                           if fwd_ev is not None else "HIP-event timers around the tagged launches of the call-by-call schedule",
                    "layer_ms": None if fwd_ev is None else [fwd_ev[3][0] / max(fwd_ev[0], 1), fwd_ev[3][1] / max(fwd_ev[0], 1)],
                    "layer1": None if fwd_ev is None else {"avg_launch_ms": fwd_ev[3][1] / max(fwd_ev[0], 1),
                                                           "algorithmic_flops_per_launch": fwd_ev[4][1] / max(fwd_ev[0], 1),
                                                           "frac": fwd_ev[4][1] / max(fwd_ev[3][1], 1e-9) / 1e9 / gemm_peak},
                    "layer_tflops": None if fwd_ev is None else [fwd_ev[4][i] / max(fwd_ev[3][i], 1e-9) / 1e9 for i in range(2)],
                    # (HBM-side bytes of THE kernel `achieved` describes: the layer-0 forward family of the committed PMC passes)
                    "traffic": (pmc_traffic()[0].get(("gemm_nt_ln_fwd_p3_bytes_per_launch" if (kinds_hl is not None and kinds_hl[0] == 3) else
                                                      "gemm_nt_p3_bytes_per_launch") if trainer._planes_on() else "gemm_nt_split_bytes_per_launch")
                                if split_mode else pmc_traffic()[0].get("gemm_nt_bytes_per_launch"))
                               if (args.in_feats, args.hidden) == (831, 256) else None,
                    "traffic_source": pmc_traffic()[1]}
        per_kernel = {}
        for tag, (n, tms, work) in kt.items():
            unit = "GB/s" if tag.startswith("spmm") or tag.startswith("batch") else "TFLOP/s"
            rate = work / (tms * 1e-3) / (1e9 if unit == "GB/s" else 1e12) if tms > 0 else 0.0
            per_kernel[tag] = {"launches": n, "total_ms": tms, "avg_ms": tms / max(n, 1), "rate": rate, "unit": unit}
        mean_nodes = int(np.mean([c for plan in node_counts(timed) for c in plan]))
        line = {
            "metric": "nodes/sec (fwd+bwd node classification) on PubLayNet page graphs",
            "value": nodes_total / elapsed, "unit": "nodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (GEMM operands split exactly into 3 bf16 pieces, fp32 accumulate)" if split_mode else "f32",
            "data": "synthetic",
            "config": {"workload": f"cfg2: {args.pages} synthetic PubLayNet-style page graphs per GPU per step "
                                   f"(~{mean_nodes} nodes, k-NN k=5 bidirected), GcnSAGE "
                                   f"{args.layers} layers F0={args.in_feats} hidden={args.hidden} classes=9, "
                                   f"CE + Adam(lr 0.01, wd 5e-4); the train loop of models/loop.py: a different batch "
                                   f"every step, assembled on the device from {args.resident_pages} resident pages per GPU "
                                   f"(shuffled epochs), eager launches",
                       "layers": args.layers, "in_feats": args.in_feats, "hidden": args.hidden,
                       "pages_per_gpu_per_step": args.pages, "global_pages_per_step": args.pages * world,
                       "resident_pages_per_gpu": args.resident_pages, "nodes_per_step_per_gpu": mean_nodes,
                       "parallelism": f"dp{world}",
                       "input_aggregate_cached": bool(kinds_hl is not None and kinds_hl[0] == 3)},
            "final_loss": final_loss, "roofline": roofline, "kernels": per_kernel, "host_page_generation_s": gen_s,
        }
        if long_run is not None:
            line["long_run"] = long_run
        # whole-step bounds of SURVEY 8(d): per node, layer l (F_l -> F_{l+1}, mean in-degree d), training mode, no recomputation
        dims = [args.in_feats] + [args.hidden] * (args.layers - 1) + [9]
        deg = float(sum(len(p.src) for p in pages)) / max(float(sum(p.num_nodes for p in pages)), 1.0)
        flops_node, bytes_node, mfma_bound, hbm_bound = step_bounds(dims, deg, gemm_peak, world)
        line["step_roofline"] = {"flops_per_node": flops_node, "bytes_per_node": bytes_node, "mean_in_degree": deg,
                                 "mfma_bound_nodes_per_s": mfma_bound, "hbm_bound_nodes_per_s": hbm_bound,
                                 "bound": "mfma" if mfma_bound < hbm_bound else "hbm",
                                 "frac": line["value"] / min(mfma_bound, hbm_bound)}
        # ... and what the step REALLY moves (FETCH_SIZE / WRITE_SIZE over every launch of a step, profiles/pmc_step.py): the headline
        # configuration in the default GEMM mode only -- the committed passes were run on that
        tpn = pmc_traffic()[0].get("step_p3_bytes_per_node")
        if tpn and (args.in_feats, args.hidden, args.layers) == (831, 256, 3) and trainer._planes_on():
            line["step_roofline"].update({"traffic_bytes_per_node": tpn, "traffic_over_algorithmic": tpn / bytes_node,
                                          "traffic_GBs": tpn * line["value"] / world / 1e9, "traffic_source": pmc_traffic()[1]})
        if extras and not args.no_replay:
            # round 1's headline mode, kept as a secondary: HIP-graph replay of 4 pre-captured resident batches
            fixed = [resident.batch(ids) for ids in timed[0][:4]]
            replays = [trainer.capture(g, g.ndata["label"]) for g in fixed]
            for i in range(8):
                replays[i % len(replays)]()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(args.steps):
                replays[i % len(replays)]()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t2
            n2 = sum(fixed[i % len(fixed)].num_nodes() for i in range(args.steps))
            line["replay"] = {"workload": "HIP-graph replay of 4 pre-captured resident batches (no batch assembly)",
                              "value": n2 / el2, "unit": "nodes/s", "ms_per_step": el2 / args.steps * 1e3}
            trainer.release()
            del fixed, replays
        if extras and not args.no_split_probe:
            line["gemm_modes"] = gemm_mode_probe(ops, run, alt, global_counts(alt), dev)
        line.update(pre)
        line["order"] = ("gather / val_graph probes, long_run (its own warm-up steps, then >= 1 s of the same loop), THEN W warm-up + K timed steps = value, "
                         "then kernel timers, replay, the other GEMM mode, inference, secondary, size_sweep, shapes, residency, cfg3, cpu_baseline")
        if extras and not args.no_inference:
            line["inference"] = inference_probe(args, gte, model, dev, pages, trainer)
        if pages13 is not None:
            line["secondary"] = secondary_probe(args, gte, S, dev, pages13)
        if extras and not args.no_size_sweep and split_mode:
            line["size_sweep"] = size_sweep_probe(args, trainer, pipe, sizes, loop, ep)
        if extras and not args.no_shapes and split_mode and pages363 is not None:
            sets = {831: pages[:300], 363: pages363}
            if pages13 is not None:
                sets[13] = pages13[:300]
            line["shapes"] = shapes_probe(args, gte, dev, sets, loop)
        if extras and not args.no_uncached and split_mode and line["config"]["input_aggregate_cached"]:
            line["uncached"] = uncached_probe(args, gte, dev, pages, loop)
        if extras and not args.no_residency and split_mode:
            line["residency"] = residency_probe(args, gte, dev, pages + (pages_res or []), loop)
            pages_res = None
        if extras and not args.no_cfg3:
            line["cfg3"] = cfg3_probe(args, gte, dev)
        if extras and not args.no_dist_probe and split_mode:
            dp = dist_one_rank_probe(args, gte, dev, resident, pipe, sizes, loop, ep)
            if dp is not None:
                line["dist_1rank"] = dp
        if world == 1 and not args.no_cpu_baseline:
            ids = timed[0][0]
            cb = cpu_baseline(args, S.concat_pages([pages[int(i)] for i in ids]), state0)
            line["cpu_baseline"] = cb
            line["gpu_over_cpu"] = line["value"] / cb["value"]
        emit(line)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
