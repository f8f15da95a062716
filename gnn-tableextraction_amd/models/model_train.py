"""Training entry point on the MI355X HIP path: ``train(data, config, name_time=None)``.

Same call, config keys, run name, checkpoint / weights / results-JSON layout as the reference's
``src/models/model_train.py:44-457`` (with ``src/utils/training.py:14-49``), restated around the HIP
engine: one flat parameter buffer, fused Adam, HIP cross-entropy, and -- when launched with
torchrun on several GPUs -- pages sharded per step with one RCCL gradient all-reduce.
What is NOT here, by design (SURVEY 8 scope): feature generation (``_generate_features`` at :242,
:293) -- graphs arrive pre-built with ``ndata['feat']``; TensorBoard is optional.
The reference falls back to the CPU when CUDA is missing (:128-130); this path refuses instead.
"""
from __future__ import annotations

import json
import os
from math import inf

import numpy as np
import torch
import torch.nn.functional as F

from .. import distributed as D
from .. import graph as G
from .. import ops
from ..components.features.utils import calculate_hidden, get_in_feats_
from ..components.graphs.models import GcnSAGE
from ..utils.config import AttrDict, logs_from_config
from ..utils.training import EarlyStopping
from .engine import FusedGcnSageStep, TrainStep
from .loop import BatchPipeline, run_steps

TABLE_TCELL, TABLE_COLH = 10, 7          # Categories_names values used for the printed F1s (const.py:4-18; KAT: tests/test_aux_golden.py)


class _ScalarLog:
    """SummaryWriter when tensorboard is installed, else a JSONL file with the same add_scalar calls."""

    def __init__(self, path):
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.w, self.f = SummaryWriter(path), None
        except Exception:
            os.makedirs(path, exist_ok=True)
            self.w, self.f = None, open(os.path.join(path, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, step):
        if self.w is not None:
            self.w.add_scalar(tag, value, step)
        else:
            self.f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
            self.f.flush()


def _class_weights(config, all_labels, n_classes):
    if not config.TRAINING.class_weights:
        return None
    method = config.TRAINING.class_weights_method
    if method == 'auto':                                   # model_train.py:106-112
        from sklearn.utils import class_weight
        wl = all_labels[all_labels != 4]
        cw = class_weight.compute_class_weight(class_weight='balanced', classes=np.unique(wl), y=wl)
        return np.insert(cw, 4, 0.1)
    if method == 'default':                                # :113-116
        return np.insert(np.asarray([1.] * 8), 6, 2.)
    raise ValueError('please specify the "class_weights_method" attribute')


def _hidden_width(config, in_feats, n_classes):
    mode = config.TRAINING.mode_params
    if mode not in ['fixed', 'scaled', 'half']:
        raise ValueError(f'Mode {mode} not in list: check config file. Exit.')
    if mode == 'fixed':
        return config.MODES.fixed.h_layer_dim
    if mode == 'scaled':
        assert config.MODES.scaled.params_no is not None and config.TRAINING.n_layers is not None
        return calculate_hidden(in_feats, n_classes, config.MODES.scaled.params_no, config.TRAINING.n_layers)
    return in_feats / 2


def adam_state_dict(step: TrainStep, model) -> dict:
    """torch.optim.Adam.state_dict() layout from the flat buffers (checkpoint compatibility, :411-419)."""
    state, off = {}, 0
    for i, p in enumerate(q for q in model.parameters() if q.requires_grad):
        n = p.numel()
        state[i] = {'step': torch.tensor(float(step.t)),
                    'exp_avg': step.exp_avg[off:off + n].view_as(p).detach().cpu().clone(),
                    'exp_avg_sq': step.exp_avg_sq[off:off + n].view_as(p).detach().cpu().clone()}
        off += n
    return {'state': state, 'param_groups': [{'lr': step.lr, 'betas': step.betas, 'eps': step.eps,
                                              'weight_decay': step.weight_decay, 'amsgrad': False,
                                              'params': list(range(len(state)))}]}


def load_adam_state_dict(step: TrainStep, model, sd: dict) -> None:
    off = 0
    for i, p in enumerate(q for q in model.parameters() if q.requires_grad):
        n = p.numel()
        st = sd['state'].get(i)
        if st is not None:
            step.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
            step.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
            step.t = int(st['step'])
        off += n
    step.lr = sd['param_groups'][0]['lr']


def evaluate(model, graph, labels, class_weights=None, engine=None):
    """no_grad forward on one (batched) graph: (loss, accuracy, predictions) -- :349-353.  ``engine`` (a FusedGcnSageStep over
    ``model``): the forward is one host call on the step's buffers (engine.forward_logits) instead of the module path."""
    model.eval()
    with torch.no_grad():
        logits = engine.forward_logits(graph) if hasattr(engine, "forward_logits") else model(graph)
        out3, _ = ops.weighted_ce(logits, labels, class_weights, want_grad=False)
        pred = logits.argmax(dim=1)
    o = out3.cpu().tolist()
    return o[0], o[2] / max(labels.shape[0], 1), pred


LAST_RUN = None          # {"model", "step", "rank", "world"} of the last train() call in this process


def _check_layout(saved, now, path):
    """A resumed run must stand on the residency layout of the run it continues: tier, ranks, windows, passes (they fix the order the
    pages are visited in).  A checkpoint written before round 6 has no layout: it resumes only onto the all-resident tier."""
    if saved is None:
        if now['tier'] != "all":
            raise RuntimeError(f"checkpoint '{path}' records no residency layout (written before round 6) and this run takes the "
                               f"'{now['tier']}' tier: the position of its page stream cannot be restored.  Set GTE_RESIDENT_BUDGET_GB so "
                               f"that the set is resident, or restart the run.")
        return
    keys = ('tier', 'world', 'passes', 'batch_size', 'ranges')
    diff = [k for k in keys if saved.get(k) != now.get(k)]
    if diff:
        raise RuntimeError(f"checkpoint '{path}' was written under another residency layout ({', '.join(f'{k}: {saved.get(k)!r}'[:80] for k in diff)}; now "
                           f"{', '.join(f'{k}: {now.get(k)!r}'[:80] for k in diff)}): the resumed page stream would not continue the interrupted one.  "
                           f"Run with the interrupted run's GTE_RESIDENT_BUDGET_GB ({saved.get('budget_gb')}), GTE_WINDOW_PASSES and world size.")


def train(data, config, name_time=None):
    rank, local_rank, world = D.env_world()
    distributed = world > 1
    n_classes = data.num_classes
    batch_size = config.TRAINING.batch_size
    say = print if rank == 0 else (lambda *a, **k: None)
    say(f"MODE: using {' '.join(config.PREPROCESS.features)} features")
    say("DATA: classes stats:\n -> n [{}]\n -> # {}\n -> % {}".format(n_classes, data.stats['numbers'],
                                                                  data.stats['percentages']))
    in_feats = get_in_feats_(config)
    feat_w = data.graphs[0].ndata['feat'].shape[1]
    if feat_w != in_feats:
        raise ValueError(f"pre-built graphs carry {feat_w} features, config asks for {in_feats}")
    h_layer_dim = _hidden_width(config, in_feats, n_classes)
    config.TRAINING.h_layer_dim = h_layer_dim
    say(f" -> f [{in_feats}]")

    all_labels = torch.cat([g.ndata['label'] for g in data.graphs]).numpy()
    cw = _class_weights(config, all_labels, n_classes)
    if not (config.TRAINING.gpu >= 0 and torch.cuda.is_available()):
        raise RuntimeError("this train loop runs on the MI355X HIP path only (no CPU fallback); "
                           "set TRAINING.gpu >= 0 on a GPU box")
    device = torch.device('cuda', local_rank if distributed else config.TRAINING.gpu)
    torch.cuda.set_device(device)
    if distributed:
        D.init_process_group("nccl", device)
    class_weights = None if cw is None else torch.tensor(cw, dtype=torch.float32, device=device)
    # optional: arithmetic of the transform GEMMs (config.TRAINING.gemm_mode = "f32" | "split_bf16"; default: GTE_GEMM_MODE / f32)
    gemm_mode = config.TRAINING.get('gemm_mode', None) if hasattr(config.TRAINING, 'get') else getattr(config.TRAINING, 'gemm_mode', None)
    if gemm_mode is not None:
        ops.set_gemm_mode(gemm_mode)
        say(f"MODE: transform GEMMs in {gemm_mode} arithmetic")

    logs = logs_from_config(config)
    out_root = config.GENERAL.get('output_dir', 'output')
    weights_dir = os.path.join(out_root, 'weights')
    ckpt_dir = os.path.join(out_root, 'checkpoints')
    res_dir = os.path.join(out_root, 'results')
    writer = _ScalarLog(os.path.join(out_root, 'runs', logs)) if rank == 0 else None
    metrics = AttrDict({'train': {'loss': inf, 'acc': 0.0}, 'val': {'loss': inf, 'acc': 0.0},
                        'f1_vect': [0.0 for _ in range(n_classes)]})

    torch.manual_seed(config.PREPROCESS.get('seed', 42))
    model = GcnSAGE(in_feats, int(h_layer_dim), n_classes, config.TRAINING.n_layers, F.relu,
                    config.TRAINING.dropout).to(device)
    say(model)
    # hand-scheduled step (no autograd, gradients written into the flat buffer) for the configuration every
    # shipped run uses (ReLU, dropout 0); anything else goes through the autograd nodes
    engine_cls = FusedGcnSageStep if not config.TRAINING.dropout else TrainStep
    step = engine_cls(model, lr=config.TRAINING.lr, weight_decay=config.TRAINING.weight_decay,
                      class_weights=class_weights, distributed=distributed)
    stopper = EarlyStopping(weights=weights_dir, name=logs, patience=config.TRAINING.es_patience)
    # ReduceLROnPlateau('min', factor=0.5) drives a stand-in optimiser whose lr is mirrored into the engine
    _lr_holder = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=config.TRAINING.lr)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(_lr_holder, 'min', factor=0.5)

    start_epoch = 0
    ck_layout = None                  # the interrupted run's residency layout (budget, tier, windows): a resumed run must reproduce it
    ckpt_path = os.path.join(ckpt_dir, logs)
    if config.GENERAL.from_checkpoint and os.path.isfile(ckpt_path):
        # our own checkpoint (written below): a dict of tensors + python scalars + the metrics dict, hence weights_only=False
        ck = torch.load(ckpt_path, map_location='cpu', weights_only=False)
        start_epoch = ck['epoch']
        model.load_state_dict(ck['state_dict'])             # parameters are views of the flat buffer: copies in place
        load_adam_state_dict(step, model, ck['optimizer'])
        metrics = AttrDict(ck['metrics'])
        ck_layout = ck.get('residency')
        _lr_holder.param_groups[0]['lr'] = step.lr
        say(f"=> loaded checkpoint '{ckpt_path}' (epoch {start_epoch})")

    train_graphs, val_graphs, _ = data.split(len(data))
    # class-weighted data parallelism: every rank knows every page's labels, hence every step's weight sums (no exchange)
    page_wsum = None
    if distributed and cw is not None:
        page_wsum = [float(np.asarray(cw, dtype=np.float64)[g.ndata['label'].long().numpy()].sum()) for g in train_graphs]
    # GTE_RESIDENT_BUDGET_GB: HBM the training pages may take per rank.  A set that fits is kept whole in HBM (below); a larger one
    # stays in pinned host memory and a window of it is resident (models/residency.py).
    # Unset: derived from the TOTAL HBM of this rank's device when the set does not fit (residency.default_budget_bytes); a resumed
    # run takes the interrupted run's budget from the checkpoint.
    from . import residency as R
    all_nodes = np.array([g.num_nodes() for g in train_graphs], dtype=np.int64)
    all_edges = np.array([g.num_edges() for g in train_graphs], dtype=np.int64)
    want_p3 = bool(getattr(step, "wants_resident_images", getattr(step, "wants_p3_features", lambda f: False))(in_feats))
    want_agg = want_p3 and bool(getattr(step, "wants_agg_image", lambda f: False)(in_feats))      # (the cached aggregate of the input)
    set_bytes = float(all_nodes.sum()) * R.WindowedPages.bytes_per_node(all_nodes, all_edges, in_feats, want_p3, want_agg)
    budget_gb = float(os.environ.get("GTE_RESIDENT_BUDGET_GB", "0") or 0)
    if budget_gb <= 0 and ck_layout is not None:
        # a resumed run takes the budget the interrupted run ran under: the budget fixes the tier and the windows, those fix the
        # order the pages are visited in (and which pages a rank owns) -- st.skip() below is only meaningful on the same layout
        budget_gb = float(ck_layout.get('budget_gb') or 0)
    elif budget_gb <= 0 and torch.device(device).type == 'cuda':
        # derived from the device's TOTAL memory, never from what happens to be free at launch (round 5 did: the data order of a
        # default run then depended on what other processes held at that moment); a device that is short of the derived budget
        # fails at allocation, loudly
        free_b, total_b = torch.cuda.mem_get_info(device)
        derived = R.default_budget_bytes(set_bytes, float(total_b))
        if derived is not None:
            budget_gb = derived / 1e9
            say(f"DATA: {set_bytes / 1e9:.1f} GB of training pages against {total_b / 1e9:.1f} GB of HBM: budget {budget_gb:.1f} GB "
                f"(set GTE_RESIDENT_BUDGET_GB to choose; {free_b / 1e9:.1f} GB are free now)")
    # Three tiers under a budget: (i) the set fits it -> all resident on every rank (the plan deals every step's pages to the
    # ranks by node count: distributed.plan_epoch); (ii) a rank's share -- the pages it OWNS, set / world -- fits -> the rank holds
    # its own pages for good and plans over them; (iii) otherwise the own pages stay in pinned host memory, a window is resident.
    tier = "all"
    if budget_gb > 0 and set_bytes > budget_gb * 1e9:
        tier = "windowed" if set_bytes / max(world, 1) > budget_gb * 1e9 else "owned"
    windowed = tier != "all"           # (both per-rank tiers run the stream loop below)
    if windowed:
        seed0 = config.PREPROCESS.get('seed', 42)
        owner = R.page_owner(len(train_graphs), world, seed0)
        # every rank's pages in a seeded random order (windows are contiguous ranges of it: unbiased samples of the set)
        rank_pages = []
        for r in range(world):
            ids = np.nonzero(owner == r)[0]
            rank_pages.append(ids[R.window_page_order(len(ids), seed0, r)])
        mine = rank_pages[rank]
        if tier == "windowed":
            passes = int(os.environ.get("GTE_WINDOW_PASSES", "8"))
            host = R.HostPages([train_graphs[i] for i in mine], device)
            wp = R.WindowedPages(host, budget_gb * 1e9, want_p3, want_agg)
        else:
            passes = 1                 # one window = every page the rank owns: a pass is a shuffled epoch over them
            wp = R.OwnedResident([train_graphs[i] for i in mine], device)
            if want_p3:
                wp.to_images(want_agg)
        # every rank's stream (pure host logic): the node counts / weight sums of a step follow without communication
        streams = []
        for r in range(world):
            ids = rank_pages[r]
            rng_ = (R.WindowedPages.layout(all_nodes[ids], all_edges[ids], in_feats, budget_gb * 1e9, want_p3, want_agg) if tier == "windowed"
                    else [(0, len(ids))])
            streams.append(R.WindowStream(rng_, batch_size, passes, seed0, rank=r))
        assert streams[rank].ranges == wp.ranges
        steps_per_epoch = min(len(ids) for ids in rank_pages) // batch_size
        if steps_per_epoch == 0:
            raise ValueError(f"{len(train_graphs)} training pages do not fill one global batch of {batch_size} pages x {world} rank(s)")
        lost = sum(len(st.never_visited()) for st in streams)
        if lost:
            say(f"DATA: {lost} training pages lie in windows smaller than one batch of {batch_size} pages and are never visited")
        if tier == "windowed":
            # what this tier changes against the reference's loop, said once (model_train.py:279-283 shuffles ALL training pages
            # every epoch and visits each once per epoch): here a rank draws its batches from ONE window of its pages at a time
            # and stays on it for `passes` shuffled passes before the next window is uploaded -- the same arithmetic per step, a
            # different sampling order (GTE_WINDOW_PASSES=1 restores one visit per page and epoch at the host link's rate)
            say(f"DATA: windowed residency -- {len(wp.ranges)} windows per rank, {passes} shuffled passes over a window per visit; "
                f"the sample order DIFFERS from the reference's epoch (model_train.py:279-283: every page once per epoch, one shuffle over "
                f"the whole set): a page is seen {passes} times while its window is resident, then not until the windows come round again; "
                f"pages never visited: {lost}.  GTE_WINDOW_PASSES=1 keeps the reference's one visit per epoch (host-link bound).")
        layout = {'budget_gb': float(budget_gb), 'tier': tier, 'world': int(world), 'passes': int(passes), 'batch_size': int(batch_size),
                  'ranges': [[[int(a), int(b)] for a, b in st.ranges] for st in streams]}
        if start_epoch:
            # a resumed run continues every rank's stream where the interrupted run stood (its position is a function of the steps
            # taken, as plan_epoch's plan is a function of the epoch) -- on the SAME layout only
            _check_layout(ck_layout, layout, ckpt_path)
            for st in streams:
                st.skip(start_epoch * steps_per_epoch)
        wp.prefetch(streams[rank].peek_window())
        pipe = BatchPipeline(wp.acquire(streams[rank].peek_window()))
        if tier == "windowed":
            pipe._bound_pages = (host.page_nodes, np.diff(host.sets["in"]["edge_off"]), np.diff(host.sets["out"]["edge_off"]))
            say(f"DATA: {set_bytes / 1e9:.3f} GB resident form, {set_bytes / max(world, 1) / 1e9:.3f} GB per rank > {budget_gb:.3f} GB budget: "
                f"host-resident, {len(wp.ranges)} windows per rank, {passes} passes per window visit, {wp.device_bytes / 1e9:.2f} GB on the device")
        else:
            say(f"DATA: {set_bytes / 1e9:.3f} GB resident form > {budget_gb:.3f} GB budget, {set_bytes / max(world, 1) / 1e9:.3f} GB per rank fits: "
                f"every rank holds the {len(mine)} pages it owns")
        resident = sizes = None
    else:
        # all training pages concatenated ONCE in HBM (features, labels, both CSRs, CSR-ordered weights); a batch is
        # four kernel launches of index arithmetic instead of dgl.batch(...).to(device) per step (:297)
        layout = {'budget_gb': float(budget_gb), 'tier': tier, 'world': int(world), 'passes': None, 'batch_size': int(batch_size), 'ranges': None}
        if start_epoch:
            _check_layout(ck_layout, layout, ckpt_path)
        resident = G.ResidentPages(train_graphs, device)
        pipe = BatchPipeline(resident)          # a step's batch is assembled on a side stream while the step before it runs
        sizes = resident.page_sizes()
        if want_p3:
            # the images now (run_steps would make them at the first epoch) so that the fp32 rows can go: batches are row maps
            resident.enable_p3(agg=want_agg)
            resident.drop_f32()
    val_shard = val_graphs[rank::world] if distributed else val_graphs
    val_graph = G.batch([g.to(device) for g in val_shard]) if val_shard else None
    val_labels = None if val_graph is None else val_graph.ndata['label']
    if val_graph is not None and hasattr(step, "attach_feature_image"):
        # the validation graph is the same graph every epoch (reference model_train.py:246): its feature image is made once and
        # the per-epoch forward runs on the planes kernels (engine.forward_logits)
        step.attach_feature_image(val_graph)
    label_ids = [v for v in range(n_classes)]

    say("\n### START TRAINING ###\n")
    train_loss = train_acc = float('nan')
    # The resident dataset leaves a large population of long-lived Python objects; a full garbage collection in the middle of
    # an epoch stalls the host for tens of ms while the step needs a launch every ~30 us (measured in bench.py: 0.62 -> 0.78-1.0
    # ms/step).  Collect once, then keep the survivors out of the collector's sight.
    import gc
    gc.collect()
    gc.freeze()
    try:
        for epoch in range(start_epoch, config.TRAINING.n_epochs):
            if windowed:
                # an epoch = the next len(train) // batch_size steps of every rank's stream (residency.WindowStream); the other
                # ranks' streams are advanced on the host for the global node counts / weight sums of those steps
                counts = np.zeros((steps_per_epoch, world), dtype=np.int64)
                wsums = np.zeros((steps_per_epoch, world), dtype=np.float64) if page_wsum is not None else None
                for r in range(world):
                    if r == rank and not distributed:
                        continue
                    st = streams[r] if r != rank else None
                    if st is None:          # this rank's own stream is consumed by run_windowed below: count on a copy
                        import copy
                        st = copy.deepcopy(streams[r])
                    k = 0
                    for w, chunk in st.take(steps_per_epoch):
                        p0 = st.ranges[w][0]
                        for ids in chunk:
                            gl = rank_pages[r][p0 + ids]
                            counts[k, r] = all_nodes[gl].sum()
                            if wsums is not None:
                                wsums[k, r] = float(np.asarray(page_wsum)[gl].sum())
                            k += 1
                scales = None if wsums is None else wsums[:, rank] / np.maximum(wsums.sum(axis=1), 1e-30)
                out3, last_n = R.run_windowed(step, pipe, wp, streams[rank], steps_per_epoch,
                                              n_global=counts.sum(axis=1) if distributed else None, loss_scale=scales)
                if out3 is not None:
                    o = out3.cpu().tolist()
                    train_loss, train_acc = o[0], o[2] / max(int(last_n), 1)
            else:
                plan = D.plan_epoch(sizes, batch_size, world, seed=config.PREPROCESS.get('seed', 42), epoch=epoch)
                counts = D.step_node_counts(plan, sizes)
                scales = None
                if page_wsum is not None:
                    wsums = D.step_weight_sums(plan, page_wsum)
                    scales = wsums[:, rank] / np.maximum(wsums.sum(axis=1), 1e-30)
                # the same loop bench.py times: models/loop.py
                out3 = run_steps(step, pipe, [ranks[rank] for ranks in plan], n_global=counts.sum(axis=1), loss_scale=scales)
                if out3 is not None:
                    o = out3.cpu().tolist()
                    train_loss, train_acc = o[0], o[2] / max(int(counts[-1][rank]), 1)

            # ---- validation on the (sharded) batched val graph -------------------------------------------
            if val_graph is not None:
                vl, va, pred = evaluate(model, val_graph, val_labels, class_weights, engine=step)
                n_val = val_labels.shape[0]
                y_true, y_pred = val_labels.long().cpu().numpy(), pred.cpu().numpy()
            else:
                vl, va, n_val, y_true, y_pred = 0.0, 0.0, 0, np.zeros(0, np.int64), np.zeros(0, np.int64)
            conf = np.zeros((n_classes, n_classes), dtype=np.float64)
            np.add.at(conf, (y_true, y_pred), 1.0)
            red = torch.tensor([vl * n_val, va * n_val, float(n_val)] + conf.reshape(-1).tolist(), dtype=torch.float64,
                               device=device)
            if distributed:
                import torch.distributed as dist
                dist.all_reduce(red)                              # one small vector: loss sum, correct, n, confusion
            red = red.cpu().numpy()
            n_tot = max(red[2], 1.0)
            val_loss, val_acc = float(red[0] / n_tot), float(red[1] / n_tot)
            conf = red[3:].reshape(n_classes, n_classes)
            tp = np.diag(conf)
            denom = conf.sum(0) + conf.sum(1)
            f1_vect = np.where(denom > 0, 2 * tp / np.maximum(denom, 1), 0.0)   # per-class F1, zero_division=0

            scheduler.step(val_loss)
            step.lr = _lr_holder.param_groups[0]['lr']
            early_stop, counter = (stopper.step(val_loss, model) if rank == 0 else (False, 0))
            if distributed:
                flag = torch.tensor([1.0 if early_stop else 0.0], device=device)
                dist.broadcast(flag, src=0)
                early_stop = bool(flag.item())
            conv = data.label_tranformer.origin_to_conv
            say(" -> Epoch {} | Train: Loss {:.4f} Acc {:.4f} | Validation: Loss {:.4f} | Accuracy {:.4f} | Cell F1 {:.4f}"
                " | Table Header F1 {:.4f}".format(epoch + 1, train_loss, train_acc, val_loss, val_acc,
                                                   f1_vect[conv[TABLE_TCELL]], f1_vect[conv[TABLE_COLH]]))
            if writer is not None:
                for tag, v in (('Loss/train', train_loss), ('Accuracy/train', train_acc), ('Loss/val', val_loss),
                               ('Accuracy/val', val_acc), ('f1/t-cell', f1_vect[conv[TABLE_TCELL]]),
                               ('f1/h-cell', f1_vect[conv[TABLE_COLH]]), ('Accuracy/counter', counter)):
                    writer.add_scalar(tag, v, epoch + 1)
            if early_stop:
                break
            if val_loss < metrics.val.loss:
                metrics['train']['loss'], metrics['train']['acc'] = train_loss, train_acc
                metrics['val']['loss'], metrics['val']['acc'] = val_loss, val_acc
                metrics['f1_vect'] = f1_vect.tolist()
            if rank == 0:
                os.makedirs(ckpt_dir, exist_ok=True)
                torch.save({'epoch': epoch + 1,
                            'state_dict': {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                            'optimizer': adam_state_dict(step, model), 'metrics': dict(metrics), 'residency': layout}, ckpt_path)
    finally:
        gc.unfreeze()          # a library entry point must not leave the caller's objects in the permanent generation

    say("\n### TRAINING ENDED ###\n")
    if os.environ.get("GTE_KEEP_LAST_RUN", "0") == "1":
        # inspection handle for tests (replicas identical); opt-in: it pins the model and the engine's GB-scale device buffers
        global LAST_RUN
        LAST_RUN = {"model": model, "step": step, "rank": rank, "world": world,
                    "tier": tier,
                    "windows": (len(wp.ranges), wp.uploaded_bytes, wp.device_bytes, len(mine)) if windowed else None}
    if rank == 0:
        os.makedirs(res_dir, exist_ok=True)
        path = os.path.join(res_dir, f'{logs}.json')
        results = {}
        if os.path.isfile(path):
            with open(path) as f:
                results = json.load(f)
        conv = data.label_tranformer.origin_to_conv
        results[logs] = {"train_loss": metrics.train.loss, "train_acc": metrics.train.acc,
                         "val_loss": metrics.val.loss, "val_acc": metrics.val.acc,
                         "cell_f1": metrics.f1_vect[conv[TABLE_TCELL]], "header_f1": metrics.f1_vect[conv[TABLE_COLH]]}
        with open(path, 'w') as f:
            json.dump(results, f, indent=4)
    return metrics


if __name__ == '__main__':
    from ..components.graphs.loader import PrebuiltPages
    from ..parsers.graphs import parse_args_ModelTrain
    cfg = parse_args_ModelTrain()
    src = os.environ.get("GTE_PAGES")
    dataset = PrebuiltPages.load(src) if src else PrebuiltPages.synthetic(cfg.TRAINING.num_graphs or 400,
                                                                          in_feats=get_in_feats_(cfg))
    train(dataset, cfg)
