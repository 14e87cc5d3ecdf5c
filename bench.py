#!/usr/bin/env python3
"""bench.py -- PCFLayer forward+backward throughput on MI355X (BASELINE.json's headline metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload: ``PCFLayer(in=64, out=64, weightnet=[12,16],
num_heads=8, guidance_feat_len=32)`` with ``USE_VI=True, BATCH_NORM=True``, train mode, one packed
cloud of N = 80 000 points per GPU, K = 16 self-neighbours; a step is one forward plus
``out.sum().backward()``.  Inputs are synthetic (seed 1: xyz ~ U[0,1)^3, unit normals, features
~ N(0,1)), resident in HBM before the timed region; the kNN that builds the neighbour table is
excluded from the step and reported separately (``knn_ms``).  With N > 1 GPUs every rank runs its own
cloud (the per-scene operator does not shard) under DistributedDataParallel: the only
collective is the gradient all-reduce over RCCL.  value = points processed by all ranks / wall time.

On one GPU the timed steps are replays of a HIP graph captured from one eager step (same kernels, same order;
``"hip_graph": true``; ``eager_ms_per_step`` is the same loop without the graph); with N > 1 the steps are eager under
DistributedDataParallel (``--graph``: replayed steps + one flat-bucket RCCL all-reduce).

Also reported on the one JSON line: ``roofline`` for the entry point with the largest device time per step (HIP events
around every entry point of eager steps; flops-based for the MFMA-bound edge graph, bytes-based for the aggregate),
``roofline_gather`` for the aggregate (north_star's gather/scatter phase), ``whole_step`` fractions, and ``cpu_baseline``
(the oracle's CPU restatement of the same layer on all host cores, plus the N = 4096 PointConv line; rank 0, 1 GPU only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'ml-pointconvformer_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

N_POINTS, K_NEI, C_FEAT, HEADS, C_MID, GUID = 80000, 16, 64, 8, 16, 32
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_*_f32)
# kernels behind the entry points that can dominate a step (prefixes of the names in the rocprofv3 summaries under profiles/)
ENTRY_KERNELS = {
    'pcf_hip_pcf_chain_forward': ('pcf_chain_kernel<', 'pcf_chain_tail_kernel<', 'chain_finalize_kernel'),
    'pcf_hip_pcf_chain_backward': ('pcf_chain_bwd_kernel<', 'chain_bwd_finalize_kernel', 'chain_bwd_reduce_kernel',
                                   'chain_bwd_combine_kernel', 'dw_reduce_kernel'),
    'pcf_hip_pcf_forward': ('agg_fwd_fx_mfma_kernel',),
    'pcf_hip_pcf_backward': ('agg_bwd_fx_mfma_kernel',),
}


def profile_counters(entry_point, shape):
    """PMC evidence for an entry point from the newest committed profiles (collected in separate rocprofv3 --pmc passes as
    MI355X_MICROARCH.md prescribes; bench.py cannot collect counters itself): SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES
    time-weighted over the entry point's kernels (profiles/r*_pmc_sq_counters.txt) and HBM bytes per launch = 2 x FETCH_SIZE +
    WRITE_SIZE summed over them (profiles/r*_pmc_traffic.json, only when it was taken at this workload's shape).  Values come
    with the file they were read from; None when no profile covers the entry point."""
    import glob
    pre = ENTRY_KERNELS.get(entry_point)
    out = {'mfma_busy_frac': None, 'mfma_busy_source': None, 'traffic': None, 'traffic_source': None}
    if not pre:
        return out
    pdir = os.path.join(ROOT, 'profiles')
    for path in sorted(glob.glob(os.path.join(pdir, 'r*_pmc_sq_counters.txt')), reverse=True):
        busy = mfma = 0.0
        try:
            for l in open(path):
                name = l[:45].strip()
                cols = l[45:].split()
                if name.startswith(pre) and len(cols) >= 3:
                    busy += float(cols[0])
                    mfma += float(cols[2])
        except (OSError, ValueError):
            continue
        if busy > 0:
            out.update(mfma_busy_frac=round(mfma / busy, 4), mfma_busy_source=os.path.relpath(path, ROOT))
            break
    for path in sorted(glob.glob(os.path.join(pdir, 'r*_pmc_traffic.json')), reverse=True):
        try:
            pmc = json.load(open(path))
            if pmc.get('shape') != shape:
                continue
            mine = [v for k, v in pmc['kernels'].items() if k.startswith(pre)]
            per_step = min([v.get('dispatches', 1) for v in mine] + [1 << 30])          # dispatches of a once-per-step kernel
            tot = [v['traffic_bytes'] * max(1, round(v.get('dispatches', per_step) / per_step)) for v in mine]
        except (OSError, ValueError, KeyError, TypeError):
            continue
        if tot:
            out.update(traffic=int(sum(tot)), traffic_source=os.path.relpath(path, ROOT))
            break
    return out

# Algorithmic HBM bytes per point of the aggregate operator, fp32 + int64 indices.
def _agg_bytes(Ci, Cm, H, K):
    fwd = 4 * K * Ci + 8 * K + 4 * K * H + 4 * K * Cm + 4 * Ci * Cm
    bwd = (4 * Ci * Cm + 4 * K * Ci + 8 * K + 4 * K * H + 4 * K * Cm) + (4 * K * Cm + 4 * K * H + 4 * K * Ci)
    return fwd, bwd


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def layer_cfg():
    return Cfg(attention_type='subtraction', BATCH_NORM=True, drop_path_rate=0., dropout_rate=0., USE_VI=True,
               USE_PE=True, PCONV_OPT=True, USE_CUDA_KERNEL=True, layer_norm_guidance=False)


def synth_cloud(n, seed):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(1, n, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(1, n, 3, generator=g), dim=-1)
    feats = torch.randn(1, n, C_FEAT, generator=g)
    return xyz, nrm, feats


def _host_cores():
    """CPU cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a 1-GPU box exposes all
    256 hardware threads in the mask but grants 16 cores of CPU time; 256 torch threads on 16 cores thrash: 56 s per
    iteration instead of 2)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    cores = min(cores, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if quota > 0:
                    cores = min(cores, max(1, -(-quota // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, cores)


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def cpu_baseline(state_dict, xyz, nrm, feats, idx, iters=5, gpu_result=None):
    """the oracle's restatement of PCFLayer fwd+bwd on ALL host cores this process may use
    (kind = "port"), 1 warm-up + `iters` >= 5 timed iterations, median.  With `gpu_result` (output, feature gradient and
    parameter gradients of one HIP step on the same tensors) the warm-up pass doubles as the parity check of the headline
    workload at its own size: -> (baseline record, parity record)."""
    from oracle import pcf_oracle as O
    cores = _host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and 'running' not in k)
          for k, v in state_dict.items()}
    P = O.Params(sd, '', True)
    f = feats.cpu().clone().requires_grad_(True)
    x, n, i = xyz.cpu(), nrm.cpu(), idx.cpu()

    def step(keep=False):
        out, _ = O.pcf_layer(P, x, f, i, n, num_heads=HEADS)
        out.sum().backward()
        res = None
        if keep:
            res = {'output': out.detach().clone(), 'feature_grad': f.grad.clone()}
            res.update({'grad:' + k: v.grad.clone() for k, v in sd.items() if v.grad is not None})
        for v in sd.values():
            v.grad = None
        f.grad = None
        return res

    t0 = time.perf_counter()
    ref = step(keep=gpu_result is not None)
    warm = time.perf_counter() - t0
    iters = max(1, min(iters, int(40.0 / max(warm, 1e-3))))        # bounded sample: at most ~40 s of CPU work
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    base = {'value': round(xyz.shape[1] / med, 1), 'unit': 'points/s', 'cores': torch.get_num_threads(),
            'kind': 'port', 'cpu': _cpu_model(),
            'sample': f'full workload N={xyz.shape[1]} K={idx.shape[2]}, 1 warm-up + {iters} timed iterations, median '
            f'{med * 1e3:.0f} ms; oracle/pcf_oracle.py:pcf_layer on torch CPU, all {cores} cores of the process CPU allowance '
            f'(affinity mask capped by the cgroup quota)'}
    if gpu_result is None:
        return base
    return base, parity_record(gpu_result, ref)


def parity_record(got, ref, tol=1e-3):
    """max |HIP - oracle| per tensor, in units of the oracle tensor's largest entry (parameter gradients that are
    analytically zero -- a bias in front of a batch-statistics BatchNorm -- hold rounding noise on every path: their unit
    is 1e-4 of the largest parameter gradient instead).  The feature gradient is additionally summarised by the share of
    rows beyond the tolerance: ReLU / LeakyReLU masks flipped by 1e-7 differences between two correct forwards move single
    rows (parity)."""
    gmax = max([float(v.abs().max()) for k, v in ref.items() if k.startswith('grad:')] + [1e-30])
    per, worst, worst_err = {}, None, -1.0
    for k, r in ref.items():
        if k not in got:
            continue
        g = got[k].detach().cpu().reshape(r.shape).float()
        unit = max(float(r.abs().max()), (1e-4 * gmax) if k.startswith('grad:') else 1e-30)
        err = float((g - r).abs().max()) / unit
        per[k] = err
        if k != 'feature_grad' and err > worst_err:
            worst, worst_err = k, err
    fg, fr = got['feature_grad'].detach().cpu().reshape(ref['feature_grad'].shape).float(), ref['feature_grad']
    row_err = (fg - fr).abs().amax(-1).reshape(-1) / max(float(fr.abs().max()), 1e-30)
    params = [v for k, v in per.items() if k.startswith('grad:')]
    return {'parity_max_rel_err': round(worst_err, 7), 'worst_tensor': worst, 'tolerance': tol,
            'output': round(per.get('output', float('nan')), 7), 'param_grads_max': round(max(params), 7) if params else None,
            'param_grad_tensors': len(params), 'feature_grad_max': round(per.get('feature_grad', float('nan')), 7),
            'feature_grad_rows_over_tolerance': int((row_err > tol).sum()), 'feature_grad_rows': int(row_err.numel()),
            'against': 'oracle/pcf_oracle.py:pcf_layer on the same tensors (the cpu_baseline warm-up pass), '
                       'parity_max_rel_err over output and parameter gradients'}


def cpu_baseline_pointconv(iters=5, n=4096, k=16):
    """BASELINE.md section 3b / BASELINE.json configs[0]: single PointConv(3 -> 32, weightnet [3, 16]), USE_VI = USE_PE =
    BATCH_NORM = False, N = 4096, K = 16, fwd + backward on the host cores (the reference's CPU-runnable case)."""
    from oracle import pcf_oracle as O
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(1, n, 3, generator=g)
    feats = torch.randn(1, n, 3, generator=g).requires_grad_(True)
    idx = torch.from_numpy(O.knn_bruteforce(xyz[0].numpy(), xyz[0].numpy(), k))[None]
    sd = {}
    for i, (a, b) in enumerate(((3, 8), (8, 8), (8, 16))):          # WeightNet 3 -> 8 -> 8 -> 16 (layers.py:127-171)
        sd[f'weightnet.mlp_convs.{i}.c.weight'] = torch.randn(b, a, generator=g) / a ** 0.5
        sd[f'weightnet.mlp_convs.{i}.c.bias'] = torch.zeros(b)
        sd[f'weightnet.mlp_convs.{i}.bn.weight'] = torch.ones(b)
        sd[f'weightnet.mlp_convs.{i}.bn.bias'] = torch.zeros(b)
    sd['linear.weight'] = torch.randn(32, 48, generator=g) / 48 ** 0.5
    sd['linear.bias'] = torch.zeros(32)
    for v in sd.values():
        v.requires_grad_(True)
    P = O.Params(sd, '', True)

    def step():
        out, _ = O.pointconv_layer(P, xyz, feats, idx, use_vi=False, use_pe=False)
        out.sum().backward()
        for v in sd.values():
            v.grad = None
        feats.grad = None

    step()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {'value': round(n / med, 1), 'unit': 'points/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'test_configs/pointconv_single.yaml layer: PointConv(3->32, weightnet [3,16]) fwd+bwd, N={n} K={k}, '
                      f'1 warm-up + {iters} timed iterations, median {med * 1e3:.1f} ms'}


LITE_GRID = [0.1, 0.2, 0.4, 0.8, 1.6]        # configs/configPCF_10cm_lite.yaml grid_size (subsample workload)


def cpu_baseline_train(cfg, net, scene_points, seed=77):
    """One training iteration of the same model through the oracle's CPU restatement (kind "port") on a BOUNDED sample:
    one synthetic scene of `scene_points` points (the GPU workload is scenes x ~40k; the oracle would need minutes for it),
    its levels and neighbour tables made by the oracles too (grid_subsample_oracle, knn_c) -- forward, cross entropy,
    backward on all host cores of the process allowance; reported as level-0 points/s so that it compares with
    `points_per_s` of the GPU line."""
    import numpy as np
    from oracle import grid_subsample_oracle as G
    from oracle import knn_c
    from oracle import pcf_oracle as O
    cores = _host_cores()
    torch.set_num_threads(cores)
    rng = np.random.default_rng(seed)
    side = cfg.grid_size[0] * scene_points ** 0.5 * 1.05
    xy = rng.random((int(scene_points * 1.6), 2), dtype=np.float32) * np.float32(side)
    z = (0.35 * np.sin(1.1 * xy[:, 0]) + 0.25 * np.cos(0.7 * xy[:, 1])).astype(np.float32)
    pts = np.stack([xy[:, 0], xy[:, 1], z], 1).astype(np.float32)
    pts, _, _ = G.grid_subsampling(pts, None, None, float(cfg.grid_size[0]))          # ~1 point per level-0 voxel
    pts = pts[:scene_points]
    nrm = np.tile(np.array([[0., 0., 1.]], np.float32), (pts.shape[0], 1))
    pcs, nrms = [pts], [nrm]
    for gs in cfg.grid_size[1:]:
        q, f, _ = G.grid_subsampling(pcs[-1], nrms[-1], None, float(gs))
        if q.shape[0] <= 16:
            q, f = pcs[-1], nrms[-1]
        pcs.append(q.astype(np.float32))
        nrms.append(f.astype(np.float32))
    K = 16
    off = [np.array([0, p.shape[0]], np.int32) for p in pcs]
    knn = lambda r, q: torch.from_numpy(knn_c.knn_packed(pcs[r], pcs[q], off[r], off[q], K).astype(np.int64))[None]
    L = len(pcs)
    es = [knn(l, l) for l in range(L)]
    ef = [knn(l, l + 1) for l in range(L - 1)]
    ep = [knn(l + 1, l) for l in range(L - 1)]
    table = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and 'running' not in k)
             for k, v in net.state_dict().items()}
    P = O.Params(table, '', True)
    feats = torch.randn(1, pts.shape[0], 3)
    target = torch.randint(0, cfg.num_classes, (pts.shape[0],))
    tp = [torch.from_numpy(p)[None] for p in pcs]
    tn = [torch.from_numpy(n)[None] for n in nrms]

    def step():
        logits = O.segmentation_model(P, cfg, feats, tp, es, ef, ep, tn)
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, cfg.num_classes), target, label_smoothing=cfg.label_smoothing)
        loss.backward()
        for v in table.values():
            v.grad = None
    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    iters = max(1, min(3, int(20.0 / max(warm, 1e-3))))
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {'value': round(pts.shape[0] / med, 1), 'unit': 'level-0 points/s', 'iters_per_s': round(1.0 / med, 4), 'cores': cores,
            'kind': 'port', 'cpu': _cpu_model(),
            'sample': f'one synthetic scene of {pts.shape[0]} level-0 points (levels {[p.shape[0] for p in pcs]}), forward + '
                      f'cross entropy + backward of the same model through oracle/pcf_oracle.py:segmentation_model, 1 warm-up + '
                      f'{iters} timed iteration(s), median {med * 1e3:.0f} ms; neighbour tables from oracle/knn_ref.c (not timed)'}


def bench_train(args):
    """Second half of BASELINE.json's metric: training iterations/s of a BASELINE model YAML (--model) on synthetic scenes;
    a step = post-kNN + inverse CSR + forward + CE loss + backward + clip_grad_norm_ + AdamW step on one packed batch per
    GPU.  One GPU: the iteration replayed from one HIP graph per pooled batch (pcf_train.GraphedTrainingStep) and, beside
    it, the eager iteration.  N > 1: pcf_train.DataParallelStep -- one flat gradient bucket, ONE RCCL all-reduce per step,
    the two halves of the iteration replayed from graphs when a short trial says replay is faster on every rank (else the
    same halves eagerly); --ddp times eager iterations under torch DistributedDataParallel instead."""
    import pcf_cuda
    import pcf_dist
    rank, world, local_rank, dev = pcf_dist.setup('nccl')
    import pcf_model
    import pcf_train
    cfg = pcf_train.baseline_config(args.model)
    if args.points is None:
        args.points = cfg.scene_points
    if args.scenes is None:
        args.scenes = cfg.scenes
    torch.manual_seed(1)
    net = pcf_model.PointConvFormer_Segmentation(cfg).to(dev).train()
    sync_bn = bool(args.sync_bn) and world > 1
    if sync_bn:
        # the reference's YAMLs set sync_bn: True and the script converts the model (train_ScanNet_DDP_WarmUP.py:192-193):
        # batch statistics over all ranks.  This build then takes every BatchNorm out of the fused chains
        # (pcf_fused.cross_rank_bn) and runs the collectives of pcf_fused.sync_bn_act inside the forward, which a HIP
        # graph cannot hold: eager halves.  Default: per-rank statistics (sync_bn False), fused chains, replay.
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    use_ddp = world > 1 and args.ddp
    use_graph = not args.no_graph and not use_ddp and not sync_bn
    opt = pcf_train.make_optimizer(cfg, net, capturable=use_graph)
    crit = torch.nn.CrossEntropyLoss(ignore_index=cfg.ignore_label, label_smoothing=cfg.label_smoothing).to(dev)
    # a small pool of distinct packed batches, rotated, so the kNN / CSR work is real every step
    pool = []
    for b in range(2):
        scenes = [pcf_train.synthetic_scene(args.points, cfg.grid_size, seed=1000 * (rank + 1) + 10 * b + i, device=dev)
                  for i in range(args.scenes)]
        pool.append(pcf_train.pack_batch(scenes, cfg.grid_size))
    n_pts = sum(pool[0][4][0])
    quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
    if quiet is not None:          # capture warm-ups run on a side stream by design (graph capture rules)
        quiet(False)

    bucket, dp, model = None, None, net
    if use_ddp:
        model = pcf_dist.wrap_ddp(net, dev)
    elif world > 1:
        bucket = pcf_dist.GradBucket(list(net.parameters()), list(net.buffers()))
        bucket.broadcast_parameters()
        dp = pcf_train.DataParallelStep(net, opt, crit, cfg, bucket, use_graph=use_graph)

    def eager(i):
        if dp is not None:
            return dp.eager(pool[i % len(pool)])
        return pcf_train.training_iteration(model, opt, crit, cfg, pool[i % len(pool)])

    for i in range(args.warmup):
        eager(i)
    # launches per eager iteration (every launch site of the library reports to the launch log)
    pcf_cuda.launch_log(True)
    eager(0)
    torch.cuda.synchronize()
    n_sites = len(pcf_cuda.read_launch_log())
    pcf_cuda.launch_log(False)

    step, graphed, note = eager, False, None
    if use_graph:
        try:
            if dp is not None:
                gstep = dp
            else:
                gstep = pcf_train.GraphedTrainingStep(net, opt, crit, cfg)
                gstep.keep_graph = True          # the captured hipGraph_t stays readable: exact node counts below
            for i in range(len(pool)):          # first call per batch: capture (+ one step)
                gstep(pool[i])
            torch.cuda.synchronize()
            graphed = gstep.use_graph if dp is not None else True
            if graphed:
                step = lambda i: gstep(pool[i % len(pool)])
                for i in range(2 * len(pool)):
                    step(i)
            elif dp is not None:
                note = 'capture failed on a rank: ' + str(getattr(dp, 'capture_error', 'on another rank'))
        except Exception as exc:       # capture is an optimisation of the host side, not a requirement
            if args.graph or world > 1:          # N > 1: ranks must not part ways silently
                raise
            note = f'HIP-graph capture of the training iteration failed ({type(exc).__name__}: {exc})'
            print('bench: ' + note + '; timing eager iterations', file=sys.stderr)
            step, graphed = eager, False
    if graphed and world > 1 and not args.graph:
        # trial: replayed against eager halves, slowest rank counts; replay has to win to be used
        trial = []
        for fn in (step, eager):
            pcf_dist.fence(dev)
            t0 = time.perf_counter()
            for i in range(4):
                fn(i)
            pcf_dist.fence(dev)
            trial.append(pcf_dist.max_over_ranks(time.perf_counter() - t0, dev))
        if trial[0] > trial[1]:
            step, graphed = eager, False
            note = f'trial: replay {trial[0] * 250:.2f} ms/step, eager {trial[1] * 250:.2f} ms/step -> eager'
    pcf_dist.fence(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    pcf_dist.fence(dev)
    elapsed = pcf_dist.max_over_ranks(time.perf_counter() - t0, dev)
    eager_ms = None
    if graphed:
        pcf_dist.fence(dev)
        t0 = time.perf_counter()
        for i in range(args.steps):
            eager(i)
        pcf_dist.fence(dev)
        eager_ms = pcf_dist.max_over_ranks(time.perf_counter() - t0, dev) / args.steps * 1e3
    nodes = None
    if graphed and dp is None:
        try:          # kernels (and other node kinds, if any) of one captured iteration
            nodes = pcf_train.graph_node_counts(next(iter(gstep.graphs.values()))[0])
        except Exception as exc:
            nodes = {'error': f'{type(exc).__name__}: {exc}'}
    if rank == 0:
        sync = None if world == 1 else ('DistributedDataParallel' if use_ddp else 'one flat-bucket all-reduce per step (pcf_dist.GradBucket)')
        path = ('HIP-graph replay per packed batch' if graphed else 'eager launches')
        if world > 1:
            path += ' under DistributedDataParallel' if use_ddp else (' (kNN .. backward + pack | all-reduce | clip + AdamW)')
        line = {
            'metric': f'{args.model} train iters/sec, synthetic scenes', 'value': round(args.steps / elapsed, 3),
            'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'points_per_s': round(world * n_pts * args.steps / elapsed, 1), 'final_loss': round(float(loss), 4),
            'hip_graph': graphed, 'eager_ms_per_step': None if eager_ms is None else round(eager_ms, 3),
            'library_launch_sites_per_iteration': n_sites, 'graph_nodes_per_iteration': nodes,
            'collective_backend': None if world == 1 else dist.get_backend(), 'world_size': world,
            'step_path': path, 'grad_sync': sync, 'note': note,
            'config': {'workload': f'{args.model} model ({sum(p.numel() for p in net.parameters())} params), '
                                   f'{args.scenes} scenes x ~{args.points} points per GPU per iteration '
                                   f'({n_pts} level-0 points, levels {pool[0][4]}), kNN + CSR + fwd + bwd + AdamW',
                       'parallelism': f'dp{world}',
                       'sync_bn': sync_bn}}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline_train(cfg, net, min(args.points, 8000))
            except Exception as exc:          # the baseline is a reported aside; the GPU number stands without it
                line['cpu_baseline'] = {'error': f'{type(exc).__name__}: {exc}'}
        print(json.dumps(_json_safe(line)), flush=True)
    pcf_dist.shutdown()


def bench_subsample(args):
    """Tertiary workload: the multi-resolution levels of one packed batch -- levels 1..4 of
    `--scenes` scenes x `--points` level-0 points by barycentre grid subsampling with the configPCF_10cm_lite grid
    sizes (datasetCommon.subsample, :384-421) -- on the GPU; a step = one knn_post_dataloader_utils.subsample_packed
    call (four pcf_hip_grid_subsample launches + the per-level count read-back).  cpu_baseline = the reference's own
    C++ (oracle/_ref, kind "reference") run per scene on one host core, as its dataloader workers do."""
    import pcf_dist
    rank, world, local_rank, dev = pcf_dist.setup('nccl')
    import knn_post_dataloader_utils as U
    import pcf_cuda
    import pcf_train
    grid = LITE_GRID
    scenes = [pcf_train.synthetic_scene(args.points, grid, seed=1000 * (rank + 1) + i, device=dev) for i in range(args.scenes)]
    xyz = torch.cat([s['xyz'] for s in scenes])
    nrm = torch.cat([s['nrm'] for s in scenes])
    counts = [int(s['xyz'].shape[0]) for s in scenes]
    n0 = sum(counts)

    def step():
        return U.subsample_packed(xyz, nrm, counts, grid)

    for _ in range(args.warmup):
        step()
    timeline = pcf_cuda.record_kernel_times(True, only=('pcf_hip_grid_subsample',))
    pcf_dist.fence(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pcs, _, stored = step()
    pcf_dist.fence(dev)
    elapsed = pcf_dist.max_over_ranks(time.perf_counter() - t0, dev)
    pcf_cuda.record_kernel_times(False)
    if rank != 0:
        pcf_dist.shutdown()
        return
    level_in = [sum(c) for c in stored[:-1]]          # points entering each of the four launches
    level_out = [sum(c) for c in stored[1:]]
    # algorithmic bytes of one launch: box 12 + key 12 read, (key 8 + index 4) written; 8 radix passes x (12 read +
    # 12 written); run heads 8 + 4, scan 3 x 4; gather of point + normal rows 24; 24 written per voxel
    alg = [n * (12 + 12 + 12 + 8 * 24 + 12 + 12 + 24) + m * 24 for n, m in zip(level_in, level_out)]
    dev_ms = [e0.elapsed_time(e1) for _, e0, e1 in timeline]
    per_step_ms = sum(dev_ms) / args.steps
    achieved = sum(alg) / (per_step_ms * 1e-3) / 1e9
    line = {
        'metric': 'multi-level grid subsampling, level-0 points/sec (4 scenes x 40k, 4 levels)',
        'value': round(world * n0 * args.steps / elapsed, 1), 'unit': 'points/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 sums / u64 keys', 'data': 'synthetic',
        'config': {'workload': f'{args.scenes} scenes x ~{args.points} level-0 points per GPU ({n0} points), grid sizes '
                               f'{grid}, levels {[sum(c) for c in stored]}', 'parallelism': f'dp{world}'},
        'roofline': {'bound': 'hbm', 'kernel': 'pcf_hip_grid_subsample (keys, rocPRIM radix sort, run heads, means)',
                     'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': None,
                     'algorithmic_bytes_per_launch': int(sum(alg) / len(alg)),
                     'avg_launch_ms': round(per_step_ms / len(alg), 4)}}
    if not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline_subsample([s['xyz'].cpu().numpy() for s in scenes],
                                                      [s['nrm'].cpu().numpy() for s in scenes], grid, n0)
    print(json.dumps(_json_safe(line)), flush=True)
    pcf_dist.shutdown()


def cpu_baseline_subsample(xyzs, nrms, grid, n0, iters=5):
    """The same batch through the reference's C++ (oracle/_ref; "reference") or, where that file did not travel, the
    oracle's numpy restatement ("port"): scene by scene, level by level, one core."""
    from oracle import grid_subsample_oracle as G
    from oracle import gridsub_ref as R
    kind = 'reference' if R.available() else 'port'
    fn = R.grid_subsampling if kind == 'reference' else (lambda p, f, l, dl: G.grid_subsampling(p, f, l, dl))

    def run():
        for p, f in zip(xyzs, nrms):
            for gs in grid[1:]:
                q, g, _ = fn(p, f, None, gs)
                if q.shape[0] > 16:
                    if kind == 'reference':            # hand the next level the canonical row order, as the GPU does
                        o = G.lex_order(q)
                        q, g = q[o], g[o]
                    p, f = q, g
    run()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {'value': round(n0 / med, 1), 'unit': 'points/s', 'cores': 1, 'kind': kind,
            'sample': f'full workload ({len(xyzs)} scenes, {n0} level-0 points, 4 levels), 1 warm-up + {iters} timed runs, '
                      f'median {med * 1e3:.1f} ms; grid_subsampling.cpp:9-110 compiled from the reference sources'
                      if kind == 'reference' else f'full workload, numpy restatement, median {med * 1e3:.1f} ms'}


def _json_safe(obj):
    """The one output line must parse as strict JSON: non-finite floats (a NaN loss, an error ratio against an all-zero
    tensor) become null."""
    if isinstance(obj, float):
        return obj if obj == obj and abs(obj) != float('inf') else None
    if isinstance(obj, dict):
        return {k: _json_safe(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_json_safe(v) for v in obj]
    return obj


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a torchrun environment: start one fresh process per GPU ourselves, as the
    reference's launcher does (run_distributed.sh:1: torch.distributed.launch --nproc_per_node), before this process has
    touched the GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <same arguments>`.  The children inherit stdout (rank 0 prints the one JSON line); the parent only waits and
    exits with the launcher's return code.  Inside a torchrun environment (WORLD_SIZE set) this is a no-op and --gpus is
    taken from the environment."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
        return
    import socket
    import subprocess
    n_visible = torch.cuda.device_count()          # counting devices does not initialise the GPU runtime
    if n_visible < args.gpus and os.environ.get('PCF_DIST_REHEARSE') != '1':          # rehearsal: every rank on cuda:0, gloo
        sys.exit(f'bench.py: --gpus {args.gpus} but only {n_visible} GPU(s) visible')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')           # dmabuf IPC: what RCCL needs on this pool
    env.setdefault('OMP_NUM_THREADS', str(max(1, _host_cores() // args.gpus)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench: launching ' + ' '.join(cmd), file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def train_in_child(args):
    """The second half of BASELINE.json's metric ("ScanNet-10cm train iters/sec") for the default line: `bench.py --workload
    train --model configPCF_10cm` in a CHILD process, started before this process has touched the GPU (a process that has
    initialised the GPU must not exec, and the two measurements should not share the card), its one JSON line parsed and the
    keys a reader needs kept.  A failure of the child is reported under "train" and leaves the headline measurement alone."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'train', '--model', 'configPCF_10cm', '--gpus', '1',
           '--steps', '10', '--warmup', '2']
    if args.no_cpu_baseline:
        cmd.append('--no-cpu-baseline')
    t0 = time.perf_counter()
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=480)
    except subprocess.TimeoutExpired:
        return {'error': 'train child: no result within 480 s'}
    rec = None
    for l in res.stdout.splitlines():
        if l.startswith('{'):
            try:
                rec = json.loads(l)
            except ValueError:
                pass
    if res.returncode != 0 or rec is None:
        return {'error': f'train child exited with {res.returncode}', 'stderr_tail': res.stderr[-400:]}
    keep = ('metric', 'value', 'unit', 'ms_per_step', 'eager_ms_per_step', 'hip_graph', 'step_path', 'points_per_s', 'final_loss',
            'library_launch_sites_per_iteration', 'graph_nodes_per_iteration', 'steps', 'warmup', 'note', 'cpu_baseline')
    out = {k: rec.get(k) for k in keep}
    out['iters_per_s_replay'] = rec.get('value') if rec.get('hip_graph') else None
    out['iters_per_s_eager'] = (round(1e3 / rec['eager_ms_per_step'], 3) if rec.get('eager_ms_per_step')
                                else (rec.get('value') if not rec.get('hip_graph') else None))
    out['workload'] = rec.get('config', {}).get('workload')
    out['measured_by'] = 'child process of this run: ' + ' '.join(cmd[1:]) + f' ({time.perf_counter() - t0:.0f} s wall)'
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--points', type=int, default=None)
    ap.add_argument('--workload', choices=['layer', 'train', 'subsample'], default='layer')
    ap.add_argument('--scenes', type=int, default=None, help='scenes per GPU per iteration (train / subsample workloads)')
    ap.add_argument('--model', default='configPCF_10cm_lite',
                    choices=['configPCF_10cm_lite', 'configPCF_10cm', 'configPCF_5cm', 'configPCF_2cm_PTF2'],
                    help='model YAML of the train workload (pcf_train.BASELINE_CONFIGS)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-train', action='store_true',
                    help='layer workload: skip the configPCF_10cm training-iteration measurement that is reported under "train"')
    ap.add_argument('--no-graph', action='store_true',
                    help='time eager steps (default on 1 GPU: HIP-graph replay of the step when capture succeeds; '
                         'the eager step is marginally host-bound -- ~190 launches in 1.6 ms -- and slows down by '
                         '5-15 %% in the first process of a fresh box)')
    ap.add_argument('--graph', action='store_true',
                    help='insist on HIP-graph replay (a failed capture is an error instead of a fall-back; with N > 1 GPUs '
                         'the replay-versus-eager trial is skipped)')
    ap.add_argument('--sync-bn', action='store_true',
                    help='train workload, N > 1: convert the model to SyncBatchNorm as the reference does with sync_bn: True '
                         '(cross-rank statistics; BatchNorms leave the fused chains, eager halves)')
    ap.add_argument('--ddp', action='store_true',
                    help='N > 1: eager steps under torch DistributedDataParallel instead of the flat gradient bucket')
    ap.add_argument('--deterministic', action='store_true',
                    help='grad_x by CSR gather-reduce (bitwise reproducible) instead of float atomics')
    args = ap.parse_args()
    self_launch(args)
    if args.workload == 'train':
        return bench_train(args)
    train_extra = None
    if args.workload == 'layer' and args.gpus == 1 and 'WORLD_SIZE' not in os.environ and not args.no_train:
        try:
            train_extra = train_in_child(args)          # before this process touches the GPU
        except Exception as exc:
            train_extra = {'error': f'{type(exc).__name__}: {exc}'}
    if args.points is None:
        args.points = N_POINTS if args.workload == 'layer' else 40000
    if args.scenes is None:
        args.scenes = 4
    if args.workload == 'subsample':
        return bench_subsample(args)

    import pcf_dist
    rank, world, local_rank = pcf_dist.env_rank()
    args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback for the HIP path)'
    rank, world, local_rank, dev = pcf_dist.setup('nccl')

    import pcf_cuda
    import pcf_layers

    torch.manual_seed(1)
    cfg = layer_cfg()
    cfg['DETERMINISTIC_BACKWARD'] = bool(args.deterministic)
    layer = pcf_layers.PCFLayer(C_FEAT, C_FEAT, cfg, weightnet=[12, C_MID], num_heads=HEADS,
                                guidance_feat_len=GUID).to(dev).train()
    model = layer          # N > 1: gradients are averaged through pcf_dist.GradBucket, or DDP when the steps run eagerly

    n = args.points
    xyz, nrm, feats = synth_cloud(n, seed=pcf_dist.data_seed(1, rank))
    xyz, nrm, feats = xyz.to(dev), nrm.to(dev), feats.to(dev).requires_grad_(True)
    off = torch.tensor([0, n], dtype=torch.int32, device=dev)
    pcf_cuda.knn_packed(xyz[0], xyz[0], off, off, K_NEI)           # warm the kernel
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx = pcf_cuda.knn_packed(xyz[0], xyz[0], off, off, K_NEI)[None].contiguous()
    torch.cuda.synchronize()
    knn_ms = (time.perf_counter() - t0) * 1e3

    # inverse CSR of the neighbour table, built once per cloud like the kNN (the training loop does it
    # per iteration next to the kNN: train_ScanNet_DDP_WarmUP.py:401); reported as csr_ms
    pcf_cuda.compute_knn_inverse(idx, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inv_n, inv_k, inv_idx = pcf_cuda.compute_knn_inverse(idx, n)
    torch.cuda.synchronize()
    csr_ms = (time.perf_counter() - t0) * 1e3

    params = list(layer.parameters())        # walking the module tree every step costs ~0.1 ms of host time
    # N > 1: one flat bucket of the 13.7 k gradient floats (packed inside the replayed graph), one RCCL all-reduce and one
    # multi-tensor copy back per step: three host calls.  DistributedDataParallel's per-step hooks need eager launches (the
    # eager step is host-bound: ~1.25 ms against 1.0 ms replayed) -- `--ddp` times that standard path instead.  Whether the
    # steps are replayed or launched eagerly is decided by a short trial below (both use the bucket; the choice is agreed
    # across the ranks), so a node where replay + collective interact badly still gets a valid, eager, number.
    use_ddp = world > 1 and (args.ddp or args.no_graph)
    bucket = pcf_dist.GradBucket(params, list(layer.buffers())) if (world > 1 and not use_ddp) else None
    if bucket is not None:
        bucket.broadcast_parameters()
    elif world > 1:
        model = pcf_dist.wrap_ddp(layer, dev)

    def forward_backward():
        out, _ = model(xyz, feats, idx, nrm, None, None, None, inv_n, inv_k, inv_idx)
        out.sum().backward()

    def sync_gradients():
        if bucket is not None:
            bucket.all_reduce()
            bucket.unpack()

    def step():
        forward_backward()
        if bucket is not None:
            bucket.pack()
        sync_gradients()
        for p in params:
            p.grad = None
        feats.grad = None

    for _ in range(args.warmup):
        step()
    eager_step = step
    graph = None
    if not args.no_graph and (world == 1 or bucket is not None):
        # HIP graph of one whole step: ~190 kernel launches, memsets and allocations replayed with one call.
        # Same kernels, same order, same work; only the host-side launch cost leaves the timed region.
        captured = True
        quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if quiet is not None:          # warm-up runs on a side stream by design (graph capture rules)
            quiet(False)
        g = torch.cuda.CUDAGraph(keep_graph=True)          # the hipGraph_t stays readable: node kinds are reported below
        tick = torch.zeros(1, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()       # no collective in flight (its watchdog polls events) while the capture runs
        try:
            # N > 1: other threads of the process (the collective library's watchdog) may touch the runtime during capture
            mode = {} if world == 1 else {'capture_error_mode': 'thread_local'}
            with torch.cuda.graph(g, **mode):
                tick.add_(1)          # a trivial first node (pcf_train._graph_preamble: keep the graph's first node off the step's kernels)
                forward_backward()
                if bucket is not None:
                    bucket.pack()
        except Exception as exc:       # capture is an optimisation of the measurement, not a requirement
            if args.graph:
                raise
            print(f'bench: HIP-graph capture failed ({type(exc).__name__}: {exc}); timing eager steps', file=sys.stderr)
            captured = False
        torch.cuda.synchronize()
        if world > 1:                  # every rank replays, or none does -- agreed before the next gradient collective
            flag = torch.tensor([1 if captured else 0], dtype=torch.int32, device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            captured = bool(flag.item())

        def replay_step():
            g.replay()
            sync_gradients()

        if captured:
            for _ in range(3):
                replay_step()
            torch.cuda.synchronize()
        if captured and world > 1 and not args.graph:
            # trial: 8 replayed against 8 eager steps, slowest rank counts; replay has to win to be used
            trial = []
            for fn in (replay_step, eager_step):
                pcf_dist.fence(dev)
                t0 = time.perf_counter()
                for _ in range(8):
                    fn()
                pcf_dist.fence(dev)
                trial.append(pcf_dist.max_over_ranks(time.perf_counter() - t0, dev))
            captured = trial[0] <= trial[1]
            if rank == 0:
                print(f'bench: trial of 8 steps: replay {trial[0] * 125:.3f} ms/step, eager {trial[1] * 125:.3f} ms/step -> '
                      f'{"replay" if captured else "eager"}', file=sys.stderr)
        if captured:
            graph, step = g, replay_step
        else:
            graph, step = None, eager_step

    fence = lambda: pcf_dist.fence(dev)

    # HIP events around the launches of the aggregate kernels only (the roofline candidates): an event
    # pair per call on all ~45 entry points of a step would cost the host more than a millisecond.
    DOMINANT = ('pcf_hip_pcf_forward', 'pcf_hip_pcf_backward', 'pcf_hip_pcf_backward_csr')
    timeline = pcf_cuda.record_kernel_times(graph is None, only=DOMINANT)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    pcf_cuda.record_kernel_times(False)
    step = eager_step
    if graph is not None:          # events cannot bracket kernels inside a graph: time them in eager steps now
        timeline = pcf_cuda.record_kernel_times(True, only=DOMINANT)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        pcf_cuda.record_kernel_times(False)
    elapsed = pcf_dist.max_over_ranks(elapsed, dev)

    # device time of the bracketed entry points inside the timed region (events on the launch stream)
    per = {}
    for name, e0, e1 in timeline:
        per.setdefault(name, []).append(e0.elapsed_time(e1))
    hip_ms = {k: sum(v) / len(v) for k, v in per.items()}
    # untimed diagnostic pass: every entry point bracketed, for the per-call table
    diag = pcf_cuda.record_kernel_times(True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pcf_cuda.record_kernel_times(False)
    per_all = {}
    for name, e0, e1 in diag:
        per_all.setdefault(name, []).append(e0.elapsed_time(e1))
    hip_ms_all = {k: sum(v) / len(v) for k, v in per_all.items()}
    hip_total_ms = sum(sum(v) for v in per_all.values()) / 3

    # eager steps (what a training loop without graph capture sees), timed the same way
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eager_step()
    fence()
    eager_ms = pcf_dist.max_over_ranks(time.perf_counter() - t0, dev) / args.steps * 1e3

    graph_nodes = None
    if graph is not None:
        try:          # node kinds of the replayed step: kernels only (the library clears buffers with its own kernels)
            import pcf_train
            graph_nodes = pcf_train.graph_node_counts(graph)
        except Exception as exc:
            graph_nodes = {'error': f'{type(exc).__name__}: {exc}'}
    if rank == 0:
        Ci = C_FEAT // 4
        fwd_b, bwd_b = _agg_bytes(Ci, C_MID, HEADS, K_NEI)
        ms_per_step = elapsed / args.steps * 1e3
        # Algorithmic work per point of the entry points that can dominate the step: bytes for the
        # HBM-bound aggregate, flops of the per-edge MLPs (mlp_conv 12 288 + guidance MLP 18 432 + WeightNet 9 216
        # = 39 936 forward, 2x backward) for the MFMA-bound edge graph.
        EDGE_FLOP_FWD = 12288 + 18432 + 9216
        cand = {
            'pcf_hip_pcf_forward': ('hbm', fwd_b, 'agg_fwd_fx_mfma_kernel'),
            'pcf_hip_pcf_backward': ('hbm', bwd_b, 'agg_bwd_fx_mfma_kernel (+ grad_x memset)'),
            'pcf_hip_pcf_backward_csr': ('hbm', bwd_b, 'agg_bwd_kernel<16,true,fx> + csr_reduce_kernel'),
            'pcf_hip_pcf_chain_forward': ('mfma', EDGE_FLOP_FWD, 'pcf_chain_kernel<1,2> + pcf_chain_tail_kernel<stats|final> + 3 finalize'),
            'pcf_hip_pcf_chain_backward': ('mfma', 2 * EDGE_FLOP_FWD, 'pcf_chain_bwd_kernel<1,2,3> + 2 finalize + reduce + combine'),
        }

        def block(name, ms):
            bound, per_point, kernels = cand[name]
            work = per_point * n
            if bound == 'hbm':
                ach = work / (ms * 1e-3) / 1e9
                b = {'bound': 'hbm', 'kernel': kernels, 'entry_point': name, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': None, 'algorithmic_bytes_per_launch': work}
            else:
                ach = work / (ms * 1e-3) / 1e12
                b = {'bound': 'mfma', 'kernel': kernels, 'entry_point': name, 'achieved': round(ach, 2), 'peak': MFMA_F32_PEAK_TF,
                     'unit': 'TFLOP/s', 'frac': round(ach / MFMA_F32_PEAK_TF, 4), 'traffic': None, 'algorithmic_flop_per_launch': work}
            b['avg_launch_ms'] = round(ms, 4)
            # counters cannot be collected inside this run: read from the committed rocprofv3 --pmc summaries, source named
            try:
                b.update(profile_counters(name, {'N': n, 'K': K_NEI, 'Ci': Ci, 'Cm': C_MID, 'H': HEADS}))
            except Exception:          # evidence files are optional
                pass
            return b

        # dominant = the bracketed entry point with the largest device time per step, over ALL entry points of the step
        per_step = {k: sum(v) / 3 for k, v in per_all.items()}
        dom = max(per_step, key=per_step.get)
        roofline = block(dom, hip_ms_all[dom]) if dom in cand else {
            'bound': 'hbm', 'kernel': dom, 'entry_point': dom, 'achieved': None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': None,
            'traffic': None, 'avg_launch_ms': round(hip_ms_all[dom], 4)}
        gather = max((k for k in ('pcf_hip_pcf_forward', 'pcf_hip_pcf_backward', 'pcf_hip_pcf_backward_csr') if k in hip_ms),
                     key=lambda k: hip_ms[k])
        roofline_gather = block(gather, hip_ms[gather])
        step_flop = 3 * 73696 * n                 # 73 696 flop/point forward, fwd+bwd counted as 3x
        step_bytes = 3 * 4312 * n                 # fused-ideal compulsory bytes, 4312 B/point forward, same 3x
        line = {
            'metric': 'PCFLayer fwd+bwd points/sec (N=80k,K=16,C=64)',
            'value': round(pcf_dist.whole_job_rate(n, args.steps, world, elapsed), 1), 'unit': 'points/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'PCFLayer(64->64, heads 8, C_mid 16, VI+BN, train) fwd+bwd, one packed cloud '
                                   f'N={n} K={K_NEI} per GPU (BASELINE configs[1]-class layer; metric shape)',
                       'points_per_gpu': n, 'K': K_NEI, 'C': C_FEAT, 'parallelism': f'dp{world}'},
            'roofline': roofline,
            'roofline_gather': roofline_gather,
            'whole_step': {'flop': step_flop, 'bytes_ideal': step_bytes,
                           'frac_mfma': round(step_flop / (ms_per_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                           'frac_hbm': round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            'hip_ms_per_call': {k: round(v, 4) for k, v in sorted(hip_ms_all.items())},
            'hip_ms_per_entry_point_per_step': {k: round(v, 4) for k, v in sorted(per_step.items())},
            'hip_ms_per_step': round(hip_total_ms, 4),
            'eager_ms_per_step': round(eager_ms, 4),
            'knn_ms': round(knn_ms, 3), 'csr_ms': round(csr_ms, 3), 'hip_graph': graph is not None, 'graph_nodes': graph_nodes,
            'step_path': ('HIP-graph replay' if graph is not None else 'eager launches') +
                         ('' if world == 1 else (' + one flat-bucket RCCL all-reduce per step' if bucket is not None
                                                 else ' under DistributedDataParallel')),
            'collective_backend': None if world == 1 else dist.get_backend(), 'world_size': world,
            'grad_sync': None if world == 1 else ('one flat-bucket all-reduce per step' if bucket is not None else 'DistributedDataParallel'),
        }
        if train_extra is not None:
            line['train'] = train_extra
        if world == 1 and not args.no_cpu_baseline:
            try:
                # one eager HIP step kept for the parity check against the oracle pass the CPU baseline runs anyway
                for p in params:
                    p.grad = None
                feats.grad = None
                out, _ = model(xyz, feats, idx, nrm, None, None, None, inv_n, inv_k, inv_idx)
                out.sum().backward()
                torch.cuda.synchronize()
                gpu_result = {'output': out.detach(), 'feature_grad': feats.grad.detach()}
                gpu_result.update({'grad:' + k: p.grad.detach() for k, p in layer.named_parameters() if p.grad is not None})
                line['cpu_baseline'], line['parity'] = cpu_baseline(layer.state_dict(), xyz, nrm, feats.detach(), idx,
                                                                    gpu_result=gpu_result)
                line['parity_max_rel_err'] = line['parity']['parity_max_rel_err']
            except Exception as exc:          # the comparison is an aside of the line: the baseline itself must still be there
                line['parity'] = {'error': f'{type(exc).__name__}: {exc}'}
                line['cpu_baseline'] = cpu_baseline(layer.state_dict(), xyz, nrm, feats.detach(), idx)
            line['cpu_baseline_pointconv_single'] = cpu_baseline_pointconv()
        print(json.dumps(_json_safe(line)), flush=True)
    pcf_dist.shutdown()


if __name__ == '__main__':
    main()
