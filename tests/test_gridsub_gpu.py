"""GPU: pcf_hip_grid_subsample (csrc/grid_subsample.hip, through pcf_cuda.grid_subsample) against the fixtures of the
reference's own C++ grid subsampling and against the oracle -- bit-exact barycentres and feature means, exact voxel
order (ascending reference voxel index per sample) -- plus packed batches, edge cases and size-independent
properties at the training-scene size."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import grid_subsample_oracle as G

pytestmark = pytest.mark.gpu
CASES = ['volume', 'surface', 'dense', 'negative', 'duplicates', 'single', 'lattice']


def _run(device, p, f, dl, counts=None):
    import pcf_cuda
    off = None
    if counts is not None:
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).to(device)
    sp, sf, sc = pcf_cuda.grid_subsample(torch.from_numpy(p).to(device), None if f is None else torch.from_numpy(f).to(device),
                                         off, dl)
    return sp.cpu().numpy(), None if sf is None else sf.cpu().numpy(), sc


@pytest.mark.parametrize('name', CASES)
def test_against_reference_fixture(device, name):
    g = {k: v.numpy() for k, v in load_golden('gridsub_' + name).items()}
    f = g['features'] if g['features'].shape[1] else None
    dl = float(g['sampleDl'])
    sp, sf, sc = _run(device, g['points'], f, dl)
    assert sc == [g['ref_points'].shape[0]]
    o = G.lex_order(sp)
    assert np.array_equal(sp[o], g['ref_points'])                      # bit-exact vs the reference's C++
    if f is not None:
        assert np.array_equal(sf[o], g['ref_features'])
    op, of, _ = G.grid_subsampling(g['points'], f, None, dl)           # and the oracle's row order, exactly
    assert np.array_equal(sp, op)
    if f is not None:
        assert np.array_equal(sf, of)


def test_packed_batch_equals_per_sample(device):
    """Five samples (one empty, one single point) in one call == the oracle on each sample alone, concatenated."""
    rng = np.random.default_rng(5)
    counts = [3000, 0, 1, 1777, 4096]
    parts_p = [(rng.random((c, 3)) * rng.uniform(1, 4) + rng.uniform(-5, 5, 3)).astype(np.float32) for c in counts]
    parts_f = [rng.standard_normal((c, 3)).astype(np.float32) for c in counts]
    sp, sf, sc = _run(device, np.concatenate(parts_p), np.concatenate(parts_f), 0.15, counts)
    want = [G.grid_subsampling(p, f, None, 0.15) for p, f in zip(parts_p, parts_f)]
    assert sc == [w[0].shape[0] for w in want]
    assert np.array_equal(sp, np.concatenate([w[0] for w in want]))
    assert np.array_equal(sf, np.concatenate([w[1] for w in want]))


def test_levels_against_reference_chain(device):
    """knn_post_dataloader_utils.subsample_packed == datasetCommon.subsample run through the reference's C++."""
    import knn_post_dataloader_utils as U
    g = {k: v.numpy() for k, v in load_golden('gridsub_levels').items()}
    grid = [float(x) for x in g['grid_size']]
    pcs, nrms, stored = U.subsample_packed(torch.from_numpy(g['points']).to(device), torch.from_numpy(g['features']).to(device),
                                           [g['points'].shape[0]], grid)
    assert len(pcs) == len(grid) and pcs[0].shape == (1, g['points'].shape[0], 3)
    for j in range(1, len(grid)):
        assert stored[j] == [g[f'level{j}_points'].shape[0]]
        assert np.array_equal(pcs[j][0].cpu().numpy(), g[f'level{j}_points']), j
        assert np.array_equal(nrms[j][0].cpu().numpy(), g[f'level{j}_features']), j


def test_levels_packed_with_small_sample(device):
    """Two samples, one so small that its coarse levels would drop to <= 16 points: that sample repeats its previous
    level (datasetCommon.py:413-414), the other one keeps subsampling."""
    import knn_post_dataloader_utils as U
    rng = np.random.default_rng(9)
    a = (rng.random((5000, 3)) * np.array([6, 6, 0.3])).astype(np.float32)
    b = (rng.random((300, 3)) * np.array([0.9, 0.9, 0.2])).astype(np.float32)
    grid = [0.1, 0.2, 0.4, 0.8]
    pcs, nrms, stored = U.subsample_packed(torch.from_numpy(np.concatenate([a, b])).to(device),
                                           torch.from_numpy(np.concatenate([a, b])).to(device), [5000, 300], grid)
    wa, wb = G.subsample(a, a, grid), G.subsample(b, b, grid)
    for j in range(len(grid)):
        assert stored[j] == [wa[0][j].shape[0], wb[0][j].shape[0]], j
        assert np.array_equal(pcs[j][0].cpu().numpy(), np.concatenate([wa[0][j], wb[0][j]])), j
        assert np.array_equal(nrms[j][0].cpu().numpy(), np.concatenate([wa[1][j], wb[1][j]])), j
    assert stored[-1][1] == stored[-2][1]                                 # the small sample stopped shrinking


def test_empty_and_errors(device):
    import pcf_cuda
    z = torch.zeros(0, 3, device=device)
    sp, sf, sc = pcf_cuda.grid_subsample(z, z, torch.zeros(1, dtype=torch.int32, device=device), 0.1)
    assert sp.shape == (0, 3) and sf.shape == (0, 3) and sc == []
    sp, sf, sc = pcf_cuda.grid_subsample(z, None, None, 0.1)
    assert sp.shape == (0, 3) and sf is None and sc == [0]
    far = torch.tensor([[0., 0., 0.], [1e6, 0., 0.]], device=device)
    with pytest.raises(RuntimeError, match='2\\^18 or more voxels'):
        pcf_cuda.grid_subsample(far, None, None, 0.01)
    with pytest.raises(ValueError):
        pcf_cuda.grid_subsample(far, None, None, 0.0)
    with pytest.raises(RuntimeError, match='must be contiguous'):
        pcf_cuda.grid_subsample(torch.zeros(3, 8, device=device)[:, :3], None, None, 0.1)


def test_full_size_properties(device):
    """4 scenes x 160k points (the raw clouds the 40k-point training scenes come from) in one call: per-sample counts
    add up; count-weighted barycentres reproduce the cloud's coordinate sums (linearity); every barycentre lies in the
    box of its sample; the voxels of a sample are distinct and ascending in the reference's linear index; and the
    result is bit-identical to the oracle."""
    import pcf_cuda
    rng = np.random.default_rng(1)
    counts = [160000] * 4
    parts = [(rng.random((c, 3)) * np.array([8, 8, 2.5]) + rng.uniform(-3, 3, 3)).astype(np.float32) for c in counts]
    p = np.concatenate(parts)
    f = rng.standard_normal(p.shape).astype(np.float32)
    dl = 0.1
    sp, sf, sc = _run(device, p, f, dl, counts)
    assert sum(sc) == sp.shape[0] == sf.shape[0]
    a = b = 0
    for pts, c_in, c_out in zip(parts, counts, sc):
        sub = sp[a:a + c_out]
        key_in, origin, nx, ny = G.voxel_keys(pts, dl)
        uniq, cnt = np.unique(key_in, return_counts=True)
        assert c_out == uniq.shape[0]
        ijk = np.floor((sub - origin) / np.float32(dl)).astype(np.int64)
        key_out = ijk[:, 0] + nx * ijk[:, 1] + nx * ny * ijk[:, 2]
        inside = key_out == uniq                       # a barycentre can round onto a voxel face; allow a handful
        assert inside.mean() > 0.999
        np.testing.assert_allclose((sub.astype(np.float64) * cnt[:, None]).sum(0), pts.astype(np.float64).sum(0), rtol=1e-5)
        assert (sub >= pts.min(0) - 1e-6).all() and (sub <= pts.max(0) + 1e-6).all()
        want = G.grid_subsampling(pts, f[b:b + c_in], None, dl)
        assert np.array_equal(sub, want[0]) and np.array_equal(sf[a:a + c_out], want[1])
        a, b = a + c_out, b + c_in
