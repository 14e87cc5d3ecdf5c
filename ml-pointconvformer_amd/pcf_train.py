"""Synthetic multi-resolution scenes and one training iteration, as the reference's loop runs it.

The dataset side of the reference (ScanNet files, augmentation, CPU voxelisation in dataloader
workers) is out of scope; what is reproduced here is the *shape* of what reaches the GPU and the
sequence of device work per iteration (``train_ScanNet_DDP_WarmUP.py:376-424``):

    post-kNN over the packed batch  (compute_knn_packed + prepare)          :382-383
    inverse CSR of every edge set   (compute_knn_inverse, PCONV_OPT)         :401
    forward, cross-entropy, backward, clip_grad_norm_(10), optimizer step   :404-424

Scenes are points on a gently folded surface (indoor scans are 2-D manifolds: each halving of the
resolution keeps about a quarter of the points), thinned to the level-0 resolution, packed over the batch
exactly like ``datasetCommon.py:348-379`` ([1, sum_i N_i, C] tensors + per-sample counts) and subsampled
level by level with the YAML's ``grid_size`` ratios by the HIP barycentre grid subsampling
(``datasetCommon.subsample`` :384-421 on the GPU).
"""
from __future__ import annotations

import math

import torch

import knn_post_dataloader_utils as knn_utils
import pcf_fused


# The model + optimisation keys of the four model YAMLs BASELINE.json names (values restated from
# configs/configPCF_10cm_lite.yaml, configPCF_10cm.yaml, configPCF_5cm.yaml, configPCF_2cm_PTF2.yaml); `scene_points` /
# `scenes` are the synthetic scene size and per-GPU batch the benchmark uses for each (BASELINE.json configs[1..4]:
# 40k-point scenes for lite, ScanNet 10 cm ~ 40k-80k, 5 cm ~ 150k, 2 cm crops of <= 120k points, BATCH_SIZE 2).
_COMMON = dict(BATCH_NORM=True, USE_XYZ=True, USE_PE=True, point_dim=3, num_level=5, base_dim=64,
               feat_dim=[64, 128, 192, 256, 384], guided_level=0, num_heads=8, resblocks_back=[0, 0, 0, 0, 0],
               K_self=[16] * 5, K_forward=[16] * 5, K_propagate=[16] * 5, num_classes=20, label_smoothing=0.2,
               adamw_decay=0.05, ignore_label=-100, dropout_rate=0., dropout_fc=0., layer_norm_guidance=False,
               sync_bn=True, use_level_1=True, drop_path_rate=0., mid_dim_back=1)
BASELINE_CONFIGS = {
    'configPCF_10cm_lite': dict(_COMMON, grid_size=[0.1, 0.2, 0.4, 0.8, 1.6], mid_dim=[4] * 5, resblocks=[0, 3, 3, 3, 3],
                                learning_rate=0.02, scene_points=40000, scenes=4),
    'configPCF_10cm': dict(_COMMON, grid_size=[0.1, 0.2, 0.4, 0.8, 1.6], mid_dim=[16] * 5, resblocks=[0, 2, 4, 6, 6],
                           learning_rate=0.02, scene_points=40000, scenes=4),
    'configPCF_5cm': dict(_COMMON, grid_size=[0.05, 0.1, 0.2, 0.4, 0.8], mid_dim=[16] * 5, resblocks=[0, 2, 4, 6, 6],
                          learning_rate=0.01, scene_points=150000, scenes=1),
    'configPCF_2cm_PTF2': dict(_COMMON, grid_size=[0.02, 0.06, 0.15, 0.375, 0.9375], mid_dim=[16] * 5, mid_dim_back=3,
                               resblocks=[0, 2, 4, 6, 6, 2], use_level_1=False, drop_path_rate=0.2, learning_rate=0.01,
                               scene_points=120000, scenes=2),
}


def baseline_config(name):
    """-> pcf_model.Config for one of the BASELINE model YAMLs, defaults filled, both kernel switches on."""
    import pcf_model
    cfg = pcf_model.Config({k: (list(v) if isinstance(v, list) else v) for k, v in BASELINE_CONFIGS[name].items()})
    pcf_model.get_default_configs(cfg, num_level=cfg.num_level, base_dim=cfg.base_dim)
    cfg.PCONV_OPT, cfg.USE_CUDA_KERNEL = True, True
    return cfg


def _voxelize_first(xyz, grid):
    """One point per occupied voxel of edge `grid` (the first in index order) -> indices, ascending: the deterministic
    mode of util/voxelize.py:44-82, which thins the raw scan to the level-0 resolution -- on the GPU
    (knn_post_dataloader_utils.voxelize -> pcf_hip_voxelize)."""
    return torch.sort(knn_utils.voxelize(xyz, grid, mode='deterministic'))[0]


def synthetic_scene(n_points, grid_sizes, seed, device, n_features=3, n_classes=20):
    """One scene at the level-0 resolution: xyz, unit normals, features and labels of about `n_points` points after
    voxelisation at grid_sizes[0].  The coarser levels are made per batch by `pack_batch`."""
    g = torch.Generator().manual_seed(seed)
    side = grid_sizes[0] * math.sqrt(n_points) * 1.05           # ~1 point per level-0 voxel of the sheet
    m = int(n_points * 1.6)
    xy = torch.rand(m, 2, generator=g) * side
    fx, fy = 1.1, 0.7
    z = 0.35 * torch.sin(fx * xy[:, 0]) + 0.25 * torch.cos(fy * xy[:, 1])
    xyz = torch.stack([xy[:, 0], xy[:, 1], z], 1).to(device)
    dzdx = 0.35 * fx * torch.cos(fx * xy[:, 0])
    dzdy = -0.25 * fy * torch.sin(fy * xy[:, 1])
    nrm = torch.nn.functional.normalize(torch.stack([-dzdx, -dzdy, torch.ones_like(dzdx)], 1), dim=1).to(device)
    keep = _voxelize_first(xyz, grid_sizes[0])[:n_points]
    n0 = keep.shape[0]
    feats = torch.randn(n0, n_features, generator=g).to(device)
    labels = torch.randint(0, n_classes, (n0,), generator=g).to(device)
    return dict(xyz=xyz[keep].contiguous(), nrm=nrm[keep].contiguous(), features=feats, labels=labels)


def pack_batch(scenes, grid_sizes):
    """Scenes -> the packed batch the collate function emits (datasetCommon.py:348-379): features [1,sumN0,C],
    pointclouds / norms lists of [1,sumN_l,3], target [sumN0], points_stored [level][sample].  The levels are the
    barycentre grid subsampling of the reference's dataloader (datasetCommon.subsample, :384-421), run on the GPU over
    the whole packed batch (knn_post_dataloader_utils.subsample_packed -> pcf_hip_grid_subsample)."""
    xyz = torch.cat([s['xyz'] for s in scenes])
    nrm = torch.cat([s['nrm'] for s in scenes])
    pointclouds, norms, points_stored = knn_utils.subsample_packed(xyz, nrm, [int(s['xyz'].shape[0]) for s in scenes],
                                                                   grid_sizes)
    features = torch.cat([s['features'] for s in scenes])[None].contiguous()
    target = torch.cat([s['labels'] for s in scenes])
    return features, pointclouds, target, norms, points_stored


def build_edges(cfg, pointclouds, points_stored):
    """post-kNN + inverse CSR for one packed batch (train_ScanNet_DDP_WarmUP.py:382-383,401)."""
    es, ef, ep = knn_utils.prepare(*knn_utils.compute_knn_packed(pointclouds, points_stored, cfg.K_self, cfg.K_forward,
                                                                 cfg.K_propagate))
    inv = (None, None, None)
    if cfg.PCONV_OPT:
        inv = knn_utils.compute_knn_inverse(pointclouds, es, ef, ep)
    return es, ef, ep, inv


def make_optimizer(cfg, model, capturable=False, fused=True):
    """AdamW as the training script builds it (train_ScanNet_DDP_WarmUP.py:237-241).  On the GPU: pcf_optim.FusedAdamW --
    gradient norm, clipping (:421) and the update over the 196 parameter tensors in seven launches with the tensor lists
    as kernel arguments (torch's multi-tensor AdamW + clip_grad_norm_: ~45 launches, 0.7 ms per iteration; its for-each
    form: ~115 launches and 5 ms of host time).  `fused=False` / CPU parameters: torch.optim.AdamW."""
    params = list(model.parameters())
    on_gpu = all(p.is_cuda for p in params)
    if fused and on_gpu:
        import pcf_optim
        return pcf_optim.FusedAdamW(params, lr=cfg.learning_rate, weight_decay=cfg.adamw_decay)
    return torch.optim.AdamW(params, lr=cfg.learning_rate, weight_decay=cfg.adamw_decay, fused=on_gpu,
                             capturable=bool(capturable and on_gpu))


@torch.no_grad()
def clip_grad_norm_(optimizer, max_norm):
    """torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) (train_ScanNet_DDP_WarmUP.py:421) over the
    optimizer's own parameter lists: same arithmetic (2-norm of the per-tensor 2-norms, coefficient clamped to 1,
    gradients scaled in place), without walking the module tree of ~850 modules every iteration."""
    grads = [p.grad for g in optimizer.param_groups for p in g['params'] if p.grad is not None]
    if not grads:
        return torch.zeros(())
    total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
    torch._foreach_mul_(grads, torch.clamp(max_norm / (total + 1e-6), max=1.0))
    return total


def forward_backward(model, criterion, cfg, batch, edges=None):
    """First half of an iteration (train_ScanNet_DDP_WarmUP.py:382-419): post-kNN, inverse CSR, forward, loss, backward.
    Leaves the gradients in `.grad`; returns the detached loss (no host sync)."""
    features, pointclouds, target, norms, points_stored = batch
    es, ef, ep, inv = edges if edges is not None else build_edges(cfg, pointclouds, points_stored)
    pred = model(features, pointclouds, es, ef, ep, norms, *inv)
    logits = pred.reshape(-1, cfg.num_classes)
    if pcf_fused.cross_entropy_supported(criterion, logits):          # the same loss, three launches instead of torch's chain
        loss = pcf_fused.cross_entropy(logits, target, criterion.ignore_index, criterion.label_smoothing)
    else:
        loss = criterion(logits, target)
    loss.backward()
    return loss.detach()


def optimizer_step(optimizer, max_grad_norm=10):
    """Second half (train_ScanNet_DDP_WarmUP.py:421-424): clip_grad_norm_(10), optimizer step, gradients dropped."""
    if hasattr(optimizer, 'last_grad_norm'):          # pcf_optim.FusedAdamW: norm, clip and update in one pass
        optimizer.step(max_grad_norm=max_grad_norm)
    else:
        clip_grad_norm_(optimizer, max_grad_norm)
        optimizer.step()
    optimizer.zero_grad(set_to_none=True)


def training_iteration(model, optimizer, criterion, cfg, batch, edges=None):
    """One optimisation step on one packed batch; returns the loss tensor (no host sync)."""
    loss = forward_backward(model, criterion, cfg, batch, edges)
    optimizer_step(optimizer)
    return loss


def _training_state(model, optimizer):
    """Every tensor an optimisation step changes in place: parameters, BatchNorm buffers, the optimizer's moments and step
    counters (torch AdamW keeps them in `state`, pcf_optim.FusedAdamW additionally in its per-group device records)."""
    seen, out = set(), []

    def add(t):
        if torch.is_tensor(t) and t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            out.append(t)
    for t in list(model.parameters()) + list(model.buffers()):
        add(t)
    for st in optimizer.state.values():
        for v in st.values():
            add(v)
    for rec in getattr(optimizer, '_recs', {}).values():
        add(rec[0])
    return out


_PREAMBLE = {}


def _graph_preamble(dev):
    """First node of every captured graph: one trivial kernel on a persistent 4-byte tensor.  The round-2 replay fault
    sat at the first work of a relaunched graph (its first node, a memset, had not taken effect before the counting sort
    behind it; DESIGN.md "graph replay fault").  The library no longer emits memset nodes; this keeps whatever a runtime
    does to a graph's FIRST node away from the iteration's own kernels as well.  Cost: one ~2 us kernel per replay."""
    t = _PREAMBLE.get(dev)
    if t is None:
        raise RuntimeError('pcf_train: _graph_preamble_init(device) must run before the capture starts')
    t.add_(1)


def _graph_preamble_init(dev):
    if dev not in _PREAMBLE:
        _PREAMBLE[dev] = torch.zeros(1, dtype=torch.int32, device=dev)


_HIP_NODE_KINDS = {0: 'kernel', 1: 'memcpy', 2: 'memset', 3: 'host', 4: 'graph', 5: 'empty', 6: 'wait_event', 7: 'event_record',
                   8: 'ext_semaphore_signal', 9: 'ext_semaphore_wait', 10: 'mem_alloc', 11: 'mem_free'}


def graph_node_counts(graph):
    """{node kind: count} of a captured torch.cuda.CUDAGraph built with keep_graph=True (hipGraphGetNodes /
    hipGraphNodeGetType through ctypes).  A captured training iteration of this package must consist of kernel nodes
    only (tests/test_zz_graph_replay_gpu.py)."""
    import ctypes
    lib = ctypes.CDLL('libamdhip64.so')
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    if lib.hipGraphGetNodes(raw, None, ctypes.byref(n)) != 0:
        raise RuntimeError('hipGraphGetNodes failed')
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    if n.value and lib.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) != 0:
        raise RuntimeError('hipGraphGetNodes failed')
    counts = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        if lib.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
            raise RuntimeError('hipGraphNodeGetType failed')
        kind = _HIP_NODE_KINDS.get(t.value, str(t.value))
        counts[kind] = counts.get(kind, 0) + 1
    return counts


class GraphedTrainingStep:
    """training_iteration captured in a HIP graph per packed batch (one graph per distinct batch object) and replayed:
    the ~2000 kernel launches of an iteration -- kNN, CSR, forward, loss, backward, clipping, AdamW -- cost one host call.

    The whole iteration is device-side (no host reads: the kNN engines are chosen from host-known sizes, offsets tables
    are cached device tensors, the optimizer is built with capturable=True so its step counters live on the device).
    A graph is tied to the shapes of its batch: this serves loops that revisit a fixed set of packed batches (the
    benchmark's rotating pool); a batch object seen for the first time is captured first.

    Every call -- the first one for a batch included -- performs EXACTLY ONE optimisation step and returns that step's
    loss: the warm-up iterations a capture needs (lazy code-object loading, persistent buffers, allocator pools) run on a
    snapshot of the training state (parameters, BatchNorm buffers, optimizer moments and counters) that is restored before
    the capture, and the captured graph is then replayed once.  Each graph owns its memory pool by default
    (`share_pool=True` lets later captures reuse the first graph's pool: correct only while graphs are replayed one at a time
    and no tensor of one graph is read after another graph ran -- the loss returned here is, so it is cloned out of the pool).
    The captured region holds kernel nodes only: the library clears buffers with its own kernels (csrc/common.hip), see
    DESIGN.md "graph replay fault"."""

    def __init__(self, model, optimizer, criterion, cfg, warmup=1, share_pool=False, max_graphs=16):
        self.model, self.optimizer, self.criterion, self.cfg, self.warmup = model, optimizer, criterion, cfg, max(1, int(warmup))
        self.share_pool, self.max_graphs = bool(share_pool), int(max_graphs)
        self.graphs = {}
        self.pool = None
        self.keep_graph = False          # tests: keep the hipGraph_t so that its nodes can be inspected

    def _capture(self, batch):
        """-> (graph, loss tensor inside the graph's pool, loss of an eager first step or None)."""
        first = None
        if not self.optimizer.state:
            # a fresh optimizer creates its moments and counters in its first step: take that step eagerly -- it is this
            # call's one optimisation step -- and capture from the state it leaves
            first = training_iteration(self.model, self.optimizer, self.criterion, self.cfg, batch).clone()
            torch.cuda.synchronize()
        _graph_preamble_init(batch[0].device)
        state = _training_state(self.model, self.optimizer)
        saved = [t.clone() for t in state]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the capture stream, on a state that is restored below
            for _ in range(self.warmup):
                training_iteration(self.model, self.optimizer, self.criterion, self.cfg, batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for t, v in zip(state, saved):
                t.copy_(v)
        del saved
        g = torch.cuda.CUDAGraph(keep_graph=True) if self.keep_graph else torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool if self.share_pool else None):
            _graph_preamble(batch[0].device)
            loss = training_iteration(self.model, self.optimizer, self.criterion, self.cfg, batch)
        if self.share_pool and self.pool is None:
            self.pool = g.pool()
        return g, loss, first

    def __call__(self, batch):
        key = id(batch)
        hit = self.graphs.get(key)
        if hit is None:
            if len(self.graphs) >= self.max_graphs:          # oldest capture out (its pool returns to the allocator)
                self.graphs.pop(next(iter(self.graphs)))
            g, loss, first = self._capture(batch)
            hit = self.graphs[key] = (g, loss, batch)          # keep the batch alive: the graph reads its tensors
            if first is not None:
                return first          # the first call on a fresh optimizer already took its one (eager) step
        if hasattr(self.optimizer, 'sync_hyperparameters'):
            self.optimizer.sync_hyperparameters()          # a scheduler's learning rate reaches the replayed kernels
        hit[0].replay()
        return hit[1].clone()


class DataParallelStep:
    """One optimisation step of data-parallel training with ONE gradient collective (the reference wraps the model in
    DistributedDataParallel, train_ScanNet_DDP_WarmUP.py:191-195; run_distributed.sh:1 starts one process per GPU):

        half 1   forward_backward on this rank's packed batch, gradients packed into the flat bucket (scaled by 1/world)
        --       all-reduce of the bucket on the compute stream: RCCL over xGMI (gloo in the CPU tests)
        half 2   gradients = views of the bucket, clip_grad_norm_(10), AdamW, gradients dropped

    Same arithmetic as DDP + the reference's loop (mean of the rank-local gradients, then clip and step on every rank).
    With `use_graph` half 1 is captured per packed batch and half 2 once, so a step costs two graph launches and one
    collective on the host instead of ~2300 launches plus DDP's per-bucket hooks (the eager iteration is host-bound:
    34 ms against 25.6 ms replayed for configPCF_10cm on one GPU).  A failed capture on ANY rank switches EVERY rank to
    eager halves for good (agreed with one MIN all-reduce before the next gradient collective).  Every call performs exactly
    one step (see GraphedTrainingStep for how the capture's warm-up is kept out of the training state).
    `sync_buffers=True` broadcasts rank 0's BatchNorm running statistics before every forward, as DDP's default
    broadcast_buffers does; they do not enter a training-mode forward, so the benchmarks leave it off."""

    def __init__(self, model, optimizer, criterion, cfg, bucket, use_graph=False, sync_buffers=False, warmup=1, max_graphs=16):
        self.model, self.optimizer, self.criterion, self.cfg, self.bucket = model, optimizer, criterion, cfg, bucket
        self.use_graph, self.sync_buffers, self.warmup, self.max_graphs = bool(use_graph), bool(sync_buffers), max(1, int(warmup)), max_graphs
        self.graphs, self.step_graph = {}, None
        self.capture_failures = 0

    # ---- the two halves, eager ----
    def _half1(self, batch, edges=None):
        loss = forward_backward(self.model, self.criterion, self.cfg, batch, edges)
        self.bucket.pack()
        for p in self.bucket.params:          # the bucket holds them now
            p.grad = None
        return loss

    def _half2(self):
        self.bucket.attach()
        optimizer_step(self.optimizer)

    def eager(self, batch, edges=None):
        if self.sync_buffers:
            self.bucket.sync_buffers()
        loss = self._half1(batch, edges)
        self.bucket.all_reduce()
        self._half2()
        return loss

    # ---- capture ----
    def _agree(self, ok, dev):
        import torch.distributed as dist
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def _capture(self, batch):
        dev = self.bucket.flat.device
        first = None
        if not self.optimizer.state:          # fresh optimizer: its first step is taken eagerly (this call's one step)
            first = self.eager(batch).clone()
            torch.cuda.synchronize()
        _graph_preamble_init(dev)
        state = _training_state(self.model, self.optimizer)
        saved = [t.clone() for t in state]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.eager(batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()          # no collective in flight while the capture runs
        with torch.no_grad():
            for t, v in zip(state, saved):
                t.copy_(v)
        del saved
        ok, g, loss, gb = True, None, None, self.step_graph
        try:
            # other threads of the process (the collective library's watchdog) may touch the runtime during capture
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                _graph_preamble(dev)
                loss = self._half1(batch)
            if gb is None:
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, capture_error_mode='thread_local'):
                    _graph_preamble(dev)
                    self._half2()
        except Exception as exc:          # capture is an optimisation of the host side, not a requirement
            ok = False
            self.capture_error = f'{type(exc).__name__}: {exc}'
            self.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        if not self._agree(ok, dev):
            self.use_graph, self.graphs, self.step_graph = False, {}, None
            self.capture_failures += 1
            return None, None, first
        self.step_graph = gb
        return g, loss, first

    def __call__(self, batch):
        if not self.use_graph:
            return self.eager(batch)
        key = id(batch)
        hit = self.graphs.get(key)
        if hit is None:
            if len(self.graphs) >= self.max_graphs:
                self.graphs.pop(next(iter(self.graphs)))
            g, loss, first = self._capture(batch)
            if g is None:          # all ranks fell back together
                return first if first is not None else self.eager(batch)
            hit = self.graphs[key] = (g, loss, batch)
            if first is not None:
                return first
        if self.sync_buffers:
            self.bucket.sync_buffers()
        if hasattr(self.optimizer, 'sync_hyperparameters'):
            self.optimizer.sync_hyperparameters()
        hit[0].replay()
        self.bucket.all_reduce()
        self.step_graph.replay()
        return hit[1].clone()
