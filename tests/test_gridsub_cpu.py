"""CPU: the oracle's restatement of the barycentre grid subsampling (oracle/grid_subsample_oracle.py) against the
fixtures the reference's own C++ produced (tests/golden/gridsub_*.npz, made by tests/golden/make_gridsub_golden.py
from /root/reference/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp compiled in the build
container) -- bit-exact -- and, where the compiled reference travelled with the snapshot (oracle/_ref), against it
directly on fresh random clouds."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import grid_subsample_oracle as G
from oracle import gridsub_ref as R

CASES = ['volume', 'surface', 'dense', 'negative', 'duplicates', 'single', 'lattice']


def _np(g):
    return {k: v.numpy() for k, v in g.items()}


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_fixture(name):
    g = _np(load_golden('gridsub_' + name))
    f = g['features'] if g['features'].shape[1] else None
    p, sf, sl = G.grid_subsampling(g['points'], f, g['labels'], float(g['sampleDl']))
    o = G.lex_order(p)
    assert np.array_equal(p[o], g['ref_points'])                       # bit-exact barycentres
    if f is not None:
        assert np.array_equal(sf[o], g['ref_features'])                # bit-exact feature means
    u = g['ref_label_unique']
    assert np.array_equal(sl[o][u], g['ref_labels'][u])                # label vote wherever upstream is specified
    # the oracle's own order is ascending voxel key (the reference's linear index)
    key, _, _, _ = G.voxel_keys(g['points'], float(g['sampleDl']))
    assert p.shape[0] == np.unique(key).shape[0]


def test_oracle_levels_match_reference_chain():
    g = _np(load_golden('gridsub_levels'))
    pts, nrm = G.subsample(g['points'], g['features'], [float(x) for x in g['grid_size']])
    assert np.array_equal(pts[0], g['points'])
    for j in range(1, len(pts)):
        assert np.array_equal(pts[j], g[f'level{j}_points']), j
        assert np.array_equal(nrm[j], g[f'level{j}_features']), j


def test_small_level_repeats_previous():
    rng = np.random.default_rng(0)
    p = (rng.random((200, 3)) * 0.5).astype(np.float32)
    pts, nrm = G.subsample(p, p.copy(), [0.1, 0.4])                     # 0.4 m voxels over a 0.5 m box: <= 8 voxels
    assert pts[1] is pts[0] and nrm[1] is nrm[0]


@pytest.mark.skipif(not R.available(), reason='oracle/_ref/libgridsub_ref.so not built (needs /root/reference)')
@pytest.mark.parametrize('n,dl,scale,fdim', [(5000, 0.1, 2.0, 3), (20000, 0.05, 3.0, 3), (40000, 0.2, 10.0, 6),
                                             (30000, 0.3, 1.0, 3), (777, 0.02, 0.5, 0)])
def test_oracle_matches_compiled_reference(n, dl, scale, fdim):
    rng = np.random.default_rng(n)
    p = (rng.random((n, 3)) * scale - 0.3 * scale).astype(np.float32)
    f = rng.standard_normal((n, fdim)).astype(np.float32) if fdim else None
    lab = rng.integers(0, 20, (n, 2)).astype(np.int32)
    a = G.grid_subsampling(p, f, lab, dl)
    b = R.grid_subsampling(p, f, lab, dl)
    oa, ob = G.lex_order(a[0]), G.lex_order(b[0])
    assert np.array_equal(a[0][oa], b[0][ob])
    if fdim:
        assert np.array_equal(a[1][oa], b[1][ob])
    u = G.label_vote_is_unique(p, lab, dl)[oa]
    assert np.array_equal(a[2][oa][u], b[2][ob][u])
