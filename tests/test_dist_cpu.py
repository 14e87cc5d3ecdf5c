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
