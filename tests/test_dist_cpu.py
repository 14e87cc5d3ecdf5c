"""CPU, gloo, world_size 2: the data-parallel plumbing bench.py and training use for N > 1 GPUs.

Each rank owns its own cloud (weak scaling; the operator does not shard inside a scene)
and DistributedDataParallel averages the parameter gradients.  The layer's HIP kernels cannot run
here, so the forward is the oracle's CPU restatement of the same PCFLayer driven by the SAME
nn.Module parameters (pcf_layers.PCFLayer builds on CPU; only its forward needs the GPU): what is
checked is that rank-local clouds + gradient all-reduce give the mean of the two single-process
gradients, and that the timing helpers agree across ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _cfg():
    return Cfg(attention_type='subtraction', BATCH_NORM=True, drop_path_rate=0., dropout_rate=0., USE_VI=True,
               USE_PE=True, PCONV_OPT=True, USE_CUDA_KERNEL=True, layer_norm_guidance=False)


def _cloud(seed, n=96, k=8):
    from oracle import pcf_oracle as O
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(1, n, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(1, n, 3, generator=g), dim=-1)
    feats = torch.randn(1, n, 32, generator=g)
    idx = torch.from_numpy(O.knn_bruteforce(xyz[0].numpy(), xyz[0].numpy(), k))[None]
    return xyz, nrm, feats, idx


class OracleDriven(torch.nn.Module):
    """pcf_layers.PCFLayer's parameters, the oracle's forward."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, xyz, feats, idx, nrm):
        from oracle import pcf_oracle as O
        table = dict(self.layer.named_parameters())
        table.update(dict(self.layer.named_buffers()))
        return O.pcf_layer(O.Params(table, '', True), xyz, feats, idx, nrm, num_heads=4)[0]


def _build():
    import pcf_layers
    torch.manual_seed(7)
    return pcf_layers.PCFLayer(32, 32, _cfg(), weightnet=[12, 4], num_heads=4, guidance_feat_len=8)


def _worker(rank, world, port, out):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import pcf_dist
    r, w, lr, dev = pcf_dist.setup('gloo')
    assert (r, w, dev.type) == (rank, world, 'cpu')
    model = pcf_dist.wrap_ddp(OracleDriven(_build()), dev)
    xyz, nrm, feats, idx = _cloud(pcf_dist.data_seed(100, rank))
    pcf_dist.fence(dev)
    model(xyz, feats, idx, nrm).sum().backward()
    pcf_dist.fence(dev)
    grads = {n: p.grad.numpy().copy() for n, p in model.module.layer.named_parameters()}   # by value
    # the same average through the flat bucket that the HIP-graph replayed steps use instead of DDP's hooks
    plain = OracleDriven(_build())
    bucket = pcf_dist.GradBucket(plain.parameters())
    bucket.broadcast_parameters()
    plain(xyz, feats, idx, nrm).sum().backward()
    bucket.pack(); bucket.all_reduce(); bucket.unpack()
    for (n, p), (_, q) in zip(plain.layer.named_parameters(), model.module.layer.named_parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-6, atol=1e-7, msg=lambda m, n=n: f'bucket vs DDP {n}: {m}')
    slow = pcf_dist.max_over_ranks(1.0 + rank, dev)
    out.put((rank, grads, slow, pcf_dist.whole_job_rate(96, 3, world, slow)))
    pcf_dist.shutdown()


@pytest.mark.timeout(300)
def test_ddp_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: the mean of the two rank-local gradients
    want = None
    for rank in range(2):
        m = OracleDriven(_build())
        xyz, nrm, feats, idx = _cloud(100 + rank)
        m(xyz, feats, idx, nrm).sum().backward()
        g = {n: p.grad for n, p in m.layer.named_parameters()}
        want = g if want is None else {n: (want[n] + g[n]) / 2 for n in g}
    for rank, grads, slow, rate in res:
        assert slow == 2.0 and rate == pytest.approx(2 * 96 * 3 / 2.0)
        for n, gexp in want.items():
            torch.testing.assert_close(torch.from_numpy(grads[n]), gexp, rtol=1e-5, atol=1e-6, msg=lambda m, n=n: f'{n}: {m}')


# ---------------------------------------------------------------------------------------------------------------
# SyncBatchNorm statistics over ranks, GradBucket buffers, and a whole-model training step under DDP
# ---------------------------------------------------------------------------------------------------------------
def _sync_bn_worker(rank, world, port, out):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    import pcf_dist
    import pcf_fused
    pcf_dist.setup('gloo')
    g = torch.Generator().manual_seed(5)
    R, C = 200, 12
    z = torch.randn(R, C, generator=g) * 2 + 3
    up = torch.randn(R, C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    cut = [0, 70, R]                                   # unequal shares
    lo, hi = cut[rank], cut[rank + 1]
    bn = torch.nn.SyncBatchNorm(C)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    bn.train()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert pcf_fused.cross_rank_bn(bn)
        zl = z[lo:hi].clone().requires_grad_(True)
        y = pcf_fused.sync_bn_act(zl, bn, pcf_fused.ACT_LEAKY)
    y.backward(up[lo:hi])
    # single-process reference on the whole batch
    ref = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        ref.weight.copy_(gamma)
        ref.bias.copy_(beta)
    zf = z.clone().requires_grad_(True)
    yf = torch.nn.functional.leaky_relu(ref(zf), 0.1)
    yf.backward(up)
    torch.testing.assert_close(y.detach(), yf.detach()[lo:hi], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(zl.grad, zf.grad[lo:hi], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
    dg, db = bn.weight.grad.clone(), bn.bias.grad.clone()      # rank-local parts: their sum is the full-batch gradient
    dist.all_reduce(dg)
    dist.all_reduce(db)
    torch.testing.assert_close(dg, ref.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db, ref.bias.grad, rtol=1e-4, atol=1e-4)
    assert int(bn.num_batches_tracked) == 1
    # eval mode: SyncBatchNorm is BatchNorm, no exchange
    bn.eval()
    assert not pcf_fused.cross_rank_bn(bn)
    # GradBucket: buffers follow rank 0 (DDP's broadcast_buffers)
    lin = torch.nn.Sequential(torch.nn.Linear(3, 3), torch.nn.BatchNorm1d(3))
    with torch.no_grad():
        lin[1].running_mean.fill_(float(rank + 1))
    bucket = pcf_dist.GradBucket(lin.parameters(), lin.buffers())
    bucket.broadcast_parameters()
    assert float(lin[1].running_mean[0]) == 1.0
    with torch.no_grad():
        lin[1].running_var.fill_(float(10 + rank))
    bucket.sync_buffers()
    assert float(lin[1].running_var[0]) == 10.0
    out.put((rank, 'ok'))
    pcf_dist.shutdown()


def _run_two(worker):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        res.append(q.get(timeout=240))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


@pytest.mark.timeout(300)
def test_sync_batchnorm_statistics_and_buffers_two_ranks():
    """pcf_fused.sync_bn_act over two gloo ranks with unequal shares == BatchNorm over the whole batch (output, input
    gradient, running statistics; parameter gradients are rank-local parts that sum to the whole); GradBucket carries
    buffers like DDP."""
    assert [r[1] for r in _run_two(_sync_bn_worker)] == ['ok', 'ok']


def _seg_cfg():
    import pcf_model
    c = pcf_model.Config(USE_PE=True, num_classes=5, PCONV_OPT=False, USE_CUDA_KERNEL=True)
    pcf_model.get_default_configs(c, num_level=3, base_dim=16)
    c.update(feat_dim=[16, 32, 48], mid_dim=[4, 4, 4], mid_dim_back=1, guided_level=0, num_heads=4,
             resblocks=[0, 1, 1], resblocks_back=[0, 0, 0])
    return c


class OracleDrivenModel(torch.nn.Module):
    """pcf_model.PointConvFormer_Segmentation's parameters and buffers, the oracle's forward."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, feats, pcs, es, ef, ep, nrms, *inv):
        from oracle import pcf_oracle as O
        table = dict(self.net.named_parameters())
        table.update(dict(self.net.named_buffers()))
        return O.segmentation_model(O.Params(table, '', True), self.net.cfg, feats, pcs, es, ef, ep, nrms)


def _seg_scene(seed, counts=(120, 60, 30), K=8):
    from oracle import pcf_oracle as O
    g = torch.Generator().manual_seed(seed)
    pcs = [torch.rand(counts[0], 3, generator=g)]
    nrms = [torch.nn.functional.normalize(torch.randn(counts[0], 3, generator=g), dim=-1)]
    for c in counts[1:]:
        sel = torch.randperm(pcs[-1].shape[0], generator=g)[:c]
        pcs.append(pcs[-1][sel])
        nrms.append(nrms[-1][sel])
    knn = lambda r, q: torch.from_numpy(O.knn_bruteforce(r.numpy(), q.numpy(), K))[None]
    es = [knn(p, p) for p in pcs]
    ef = [knn(pcs[l], pcs[l + 1]) for l in range(len(pcs) - 1)]
    ep = [knn(pcs[l + 1], pcs[l]) for l in range(len(pcs) - 1)]
    feats = torch.randn(1, counts[0], 3, generator=g)
    target = torch.randint(0, 5, (counts[0],), generator=g)
    return feats, [p[None] for p in pcs], es, ef, ep, [n[None] for n in nrms], target


def _seg_model():
    import pcf_model
    torch.manual_seed(3)
    return pcf_model.PointConvFormer_Segmentation(_seg_cfg())


def _seg_step(model, scene):
    feats, pcs, es, ef, ep, nrms, target = scene
    loss = torch.nn.functional.cross_entropy(model(feats, pcs, es, ef, ep, nrms).reshape(-1, 5), target, label_smoothing=0.2)
    loss.backward()
    return loss


def _model_worker(rank, world, port, out):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import pcf_dist
    import pcf_train
    _, _, _, dev = pcf_dist.setup('gloo')
    net = _seg_model()
    model = pcf_dist.wrap_ddp(OracleDrivenModel(net), dev)
    _seg_step(model, _seg_scene(pcf_dist.data_seed(200, rank)))
    grads = {n: p.grad.numpy().copy() for n, p in net.named_parameters()}
    # the optimiser step of the training loop on the averaged gradients: every rank ends with the same parameters
    opt = torch.optim.AdamW(net.parameters(), lr=0.02, weight_decay=0.05)
    pcf_train.clip_grad_norm_(opt, 10)
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    torch.distributed.all_gather(both, flat)
    assert torch.equal(both[0], both[1])
    out.put((rank, grads))
    pcf_dist.shutdown()


@pytest.mark.timeout(300)
def test_segmentation_model_training_step_ddp_two_ranks():
    """One training step of the whole segmentation graph (pcf_model's parameters, the oracle's forward) under DDP over
    two gloo ranks with rank-local scenes: the gradients every rank ends up with are the mean of the two
    single-process gradients, and the optimiser leaves both ranks with identical parameters."""
    res = _run_two(_model_worker)
    want = None
    for rank in range(2):
        net = _seg_model()
        _seg_step(OracleDrivenModel(net), _seg_scene(200 + rank))
        g = {n: p.grad for n, p in net.named_parameters()}
        want = g if want is None else {n: (want[n] + g[n]) / 2 for n in g}
    for rank, grads in res:
        for n, gexp in want.items():
            torch.testing.assert_close(torch.from_numpy(grads[n]), gexp, rtol=1e-4, atol=1e-5 * max(1.0, float(gexp.abs().max())),
                                       msg=lambda m, n=n: f'{n}: {m}')


def _split_step_worker(rank, world, port, out):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import pcf_dist
    import pcf_train
    _, _, _, dev = pcf_dist.setup('gloo')
    cfg = _seg_cfg()
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.2)

    def batches():
        res = []
        for b in range(2):
            feats, pcs, es, ef, ep, nrms, target = _seg_scene(pcf_dist.data_seed(300 + 10 * b, rank))
            res.append(((feats, pcs, target, nrms, None), (es, ef, ep, (None, None, None))))
        return res

    # reference: DistributedDataParallel + the training loop's step
    net_a = _seg_model()
    ddp = pcf_dist.wrap_ddp(OracleDrivenModel(net_a), dev)
    opt_a = torch.optim.SGD(net_a.parameters(), lr=0.1, momentum=0.9)
    # this build: flat bucket, one all-reduce, gradients attached as views of the bucket
    net_b = _seg_model()
    bucket = pcf_dist.GradBucket(list(net_b.parameters()), list(net_b.buffers()))
    bucket.broadcast_parameters()
    opt_b = torch.optim.SGD(net_b.parameters(), lr=0.1, momentum=0.9)
    step = pcf_train.DataParallelStep(OracleDrivenModel(net_b), opt_b, crit, cfg, bucket, use_graph=False, sync_buffers=True)
    losses = []
    for it, (batch, edges) in enumerate(batches() + batches()[:1]):
        la = pcf_train.training_iteration(ddp, opt_a, crit, cfg, batch, edges)
        lb = step.eager(batch, edges)
        losses.append((float(la), float(lb)))
        assert all(p.grad is None for p in net_b.parameters()), 'gradients are dropped after the step'
    pa = torch.cat([p.detach().reshape(-1) for p in net_a.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in net_b.parameters()])
    ba = {n: b.clone() for n, b in net_a.named_buffers()}
    bb = {n: b.clone() for n, b in net_b.named_buffers()}
    both = [torch.empty_like(pb) for _ in range(world)]
    torch.distributed.all_gather(both, pb)
    assert torch.equal(both[0], both[1]), 'ranks diverged'
    out.put((rank, losses, float((pa - pb).abs().max()), float(pa.abs().max()),
             max(float((ba[n].double() - bb[n].double()).abs().max()) for n in ba)))
    pcf_dist.shutdown()


@pytest.mark.timeout(600)
def test_data_parallel_step_equals_ddp_two_ranks():
    """pcf_train.DataParallelStep (flat bucket packed after backward, ONE all-reduce, gradients attached as bucket views,
    clip + optimizer step; BatchNorm buffers broadcast from rank 0 like DDP's broadcast_buffers) against
    DistributedDataParallel + pcf_train.training_iteration over three steps on two gloo ranks with rank-local scenes:
    same losses, same parameters and buffers after every step's worth of updates, identical parameters on both ranks.
    This is the eager form of what bench.py --workload train runs with N > 1 (its two halves are what gets captured)."""
    for rank, losses, diff, scale, bdiff in _run_two(_split_step_worker):
        for la, lb in losses:
            assert abs(la - lb) < 1e-5 * max(1.0, abs(la)), (rank, losses)
        assert diff < 2e-5 * max(1.0, scale), (rank, diff, scale)
        assert bdiff < 1e-5, (rank, bdiff)


def _bench_dry_run_two_ranks(extra):
    """bench.py under `python -m torch.distributed.run --nproc-per-node 2` with tools/bench_dry_run.py's stubs: no-op
    launches, dummy graph / stream objects, REAL ranks and collectives over gloo."""
    import json
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tools', 'bench_dry_run.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1'] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'rank 0 prints exactly one JSON line'
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_train_workload_two_ranks_control_flow():
    """The N > 1 train path of bench.py end to end on two gloo ranks (kernels stubbed): parameter broadcast, DataParallelStep
    captures on both ranks, the capture agreement, the replay-versus-eager trial, the timed loop with one all-reduce per step,
    max-over-ranks timing -- no deadlock, one JSON line, whole-job keys."""
    line = _bench_dry_run_two_ranks(['--workload', 'train', '--model', 'configPCF_10cm_lite', '--points', '700', '--scenes', '2'])
    assert line['n_gpus'] == 2 and line['world_size'] == 2 and line['collective_backend'] == 'gloo'
    assert line['grad_sync'].startswith('one flat-bucket all-reduce') and line['config']['parallelism'] == 'dp2'
    assert line['scaling'] == 'weak' and line['config']['sync_bn'] is False


@pytest.mark.timeout(600)
def test_bench_layer_workload_two_ranks_control_flow():
    """The N > 1 layer path (HIP-graph replay + flat bucket + trial, or eager) on two gloo ranks with stubbed kernels."""
    line = _bench_dry_run_two_ranks(['--points', '1200'])
    assert line['n_gpus'] == 2 and line['grad_sync'] == 'one flat-bucket all-reduce per step'
    assert line['value'] > 0 and 'roofline' in line
