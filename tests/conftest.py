import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'ml-pointconvformer_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """-> dict of torch tensors (ints stay int64, floats fp32) from tests/golden/<name>.npz."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def split(blobs, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in blobs.items() if k.startswith(prefix)}


def synthetic_parameters(shapes, seed=7):
    """{name: shape} -> {name: tensor}: parameters as a deterministic function of (name, shape, seed), so that the
    whole-model fixtures need not store 5 M parameters -- the generator applies this to the reference model, the tests
    to the build's model (same parameter names).  Linear weights U(+-1/sqrt(fan_in)), BatchNorm / LayerNorm gains
    U(0.5, 1.5), biases U(+-0.1)."""
    import zlib
    out = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(name.encode()))
        u = torch.rand(tuple(shape), generator=g)
        if len(shape) >= 2:
            out[name] = (2 * u - 1) / float(shape[-1]) ** 0.5
        elif name.endswith('weight'):          # 1-D weight: a normalisation gain
            out[name] = 0.5 + u
        else:
            out[name] = 0.2 * u - 0.1
    return out


def block_inputs(name, draw, n_in, c_in, n_out, c_out, with_skip):
    """Synthetic feature input [1,n_in,c_in], decoder skip [1,n_out,c_out] (or None) and upstream gradient
    [1,n_out,c_out] of one block of a whole-model fixture, as a function of (block name, draw index)."""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) * 1009 + draw)
    feats = torch.randn(1, n_in, c_in, generator=g)
    skip = torch.randn(1, n_out, c_out, generator=g) if with_skip else None
    up = torch.randn(1, n_out, c_out, generator=g)
    return feats, skip, up


GRAD_FULL_MAX, GRAD_SAMPLE = 8192, 4096          # whole-model parameter gradients
BLOCK_FULL_MAX, BLOCK_SAMPLE = 2048, 1024          # per-block tensors of the whole-model fixtures


def grad_digest(name, grad, full_max=GRAD_FULL_MAX, sample=GRAD_SAMPLE):
    """What a whole-model fixture keeps of one tensor: all of it up to `full_max` elements, else a strided sample of
    `sample` elements plus (l2 norm, sum)."""
    g = grad.detach().reshape(-1).to(torch.float32).cpu()
    if g.numel() <= full_max:
        return {'gsd.' + name: g.reshape(grad.shape).numpy()}
    stride = g.numel() // sample
    return {'gsmp.' + name: g[::stride][:sample].numpy(),
            'gnrm.' + name: np.asarray([float(g.double().norm()), float(g.double().sum())])}


def digest_mismatch(got, blobs, prefix, name, rtol, atol, full_max=GRAD_FULL_MAX, sample=GRAD_SAMPLE):
    """Compare tensor `got` with the digest stored under `prefix` (e.g. '' or 'blk.') for `name`; -> list of problems
    (empty = equal within rtol / atol * max(1, |ref|_max))."""
    flat = got.detach().reshape(-1).float().cpu()
    probs = []
    if prefix + 'gsd.' + name in blobs:
        ref = blobs[prefix + 'gsd.' + name].reshape(-1)
        cmp = flat
    else:
        ref = blobs[prefix + 'gsmp.' + name]
        assert flat.numel() > full_max, name
        cmp = flat[::flat.numel() // sample][:sample]
        nrm = blobs[prefix + 'gnrm.' + name]
        if abs(float(flat.double().norm()) - float(nrm[0])) > 2 * rtol * max(1.0, float(nrm[0])):
            probs.append((name + ' (l2 norm)', float(flat.double().norm()), float(nrm[0])))
    scale = max(1.0, float(ref.abs().max()))
    if cmp.shape != ref.shape or not torch.allclose(cmp, ref, rtol=rtol, atol=atol * scale):
        probs.append((name, float((cmp - ref).abs().max()) if cmp.shape == ref.shape else 'shape', scale))
    return probs


@pytest.fixture(scope='session')
def device():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
