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


def training_iteration(model, optimizer, criterion, cfg, batch, edges=None):
    """One optimisation step on one packed batch; returns the loss tensor (no host sync)."""
    features, pointclouds, target, norms, points_stored = batch
    es, ef, ep, inv = edges if edges is not None else build_edges(cfg, pointclouds, points_stored)
    pred = model(features, pointclouds, es, ef, ep, norms, *inv)
    logits = pred.reshape(-1, cfg.num_classes)
    if pcf_fused.cross_entropy_supported(criterion, logits):          # the same loss, three launches instead of torch's chain
        loss = pcf_fused.cross_entropy(logits, target, criterion.ignore_index, criterion.label_smoothing)
    else:
        loss = criterion(logits, target)
    loss.backward()
    if hasattr(optimizer, 'last_grad_norm'):          # pcf_optim.FusedAdamW: norm, clip and update in one pass
        optimizer.step(max_grad_norm=10)
    else:
        clip_grad_norm_(optimizer, 10)
        optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return loss.detach()


class GraphedTrainingStep:
    """training_iteration captured in a HIP graph per packed batch (one graph per distinct batch object) and replayed:
    the ~2000 kernel launches of an iteration -- kNN, CSR, forward, loss, backward, clipping, AdamW -- cost one host call.

    The whole iteration is device-side (no host reads: the kNN engines are chosen from host-known sizes, offsets tables
    are cached device tensors, the optimizer is built with capturable=True so its step counters live on the device).
    A graph is tied to the shapes of its batch: this serves loops that revisit a fixed set of packed batches (the
    benchmark's rotating pool); batches of new shapes fall back to a fresh capture.  Replaying changes parameters,
    optimizer state and BatchNorm running statistics exactly as the eager iteration does."""

    def __init__(self, model, optimizer, criterion, cfg, warmup=2):
        self.model, self.optimizer, self.criterion, self.cfg, self.warmup = model, optimizer, criterion, cfg, warmup
        self.graphs = {}
        self.pool = None

    def _capture(self, batch):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the capture: allocator pools, lazy kernel loading, persistent buffers
            for _ in range(self.warmup):
                training_iteration(self.model, self.optimizer, self.criterion, self.cfg, batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool):
            loss = training_iteration(self.model, self.optimizer, self.criterion, self.cfg, batch)
        if self.pool is None:
            self.pool = g.pool()
        return g, loss

    def __call__(self, batch):
        key = id(batch)
        hit = self.graphs.get(key)
        if hit is None:
            hit = self.graphs[key] = self._capture(batch) + (batch,)       # keep the batch alive: the graph reads its tensors
            return hit[1]
        if hasattr(self.optimizer, 'sync_hyperparameters'):
            self.optimizer.sync_hyperparameters()          # a scheduler's learning rate reaches the replayed kernels
        hit[0].replay()
        return hit[1]
