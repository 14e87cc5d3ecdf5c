"""CPU: the voxelisation oracle against fixtures of the reference's util/voxelize.py (keys bit-exact; same voxel sequence;
'multiple' sets cover every point)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import voxelize_oracle as V

CASES = ['vox_surface', 'vox_dense', 'vox_negative', 'vox_2cm', 'vox_single', 'vox_f64_faces']


def load(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference(name):
    g = load(name)
    key = V.fnv_keys(g['coord'], float(g['voxel']))
    assert key.dtype == np.uint64 and np.array_equal(key, g['key'])
    idx, counts = V.voxelize(g['coord'], float(g['voxel']))
    assert idx.shape == g['idx'].shape
    assert np.array_equal(key[idx], g['key'][g['idx']])            # the same voxel at every output position
    ks = key[idx]
    assert np.all(ks[1:] > ks[:-1])                                # ascending key order, one entry per voxel
    first = {}
    for i, k in enumerate(key.tolist()):
        first.setdefault(k, i)
    assert all(first[k] == i for k, i in zip(key[idx].tolist(), idx.tolist()))     # lowest index of its voxel
    assert counts.sum() == g['coord'].shape[0]


def test_multiple_mode_sets():
    g = load('vox_dense')
    n_sets = sum(1 for k in g if k.startswith('multi'))
    _, counts = V.voxelize(g['coord'], float(g['voxel']))
    assert n_sets == counts.max()
    seen = set()
    for r in range(n_sets):
        idx, _ = V.voxelize(g['coord'], float(g['voxel']), mode='rank', rank=r)
        assert np.array_equal(g['key'][idx], g['key'][g[f'multi{r}']])      # same voxel sequence as the reference's set r
        seen.update(idx.tolist())
    assert len(seen) == g['coord'].shape[0]                                    # together they cover every point


def test_float64_coordinates_are_hashed_as_doubles():
    """vox_f64_faces: float64 coordinates within 1e-9 .. 1e-12 of voxel faces.  The reference divides the doubles
    themselves (any NumPy); hashing their float32 roundings would move thousands of these points into a neighbouring voxel --
    which is why the library has a float64 entry point (pcf_hip_voxelize_f64) instead of casting on the way in."""
    g = load('vox_f64_faces')
    assert g['coord'].dtype == np.float64
    assert np.array_equal(V.fnv_keys(g['coord'], float(g['voxel'])), g['key'])
    rounded = V.fnv_keys(g['coord'].astype(np.float32), float(g['voxel']))
    assert (rounded != g['key']).sum() > 1000
