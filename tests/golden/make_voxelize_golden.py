"""Golden vectors for the level-0 voxelisation from the reference's own function (util/voxelize.py:44-82, importable as
it is: numpy + torch only).  Run in the build container:  python tests/golden/make_voxelize_golden.py

Stored per case: the coordinates, the voxel size, the reference's FNV keys per point (fnv_hash_vec of the floored
quotient), its 'deterministic' selection (idx_unique, in ascending key order) and, for one case, the index sets of the
'multiple' mode."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
from util.voxelize import fnv_hash_vec, voxelize      # noqa: E402

rng = np.random.default_rng(5)
cases = {
    'vox_surface': ((rng.random((20000, 3)) * np.array([8.0, 6.0, 0.3])).astype(np.float32), 0.1),
    'vox_dense': ((rng.random((6000, 3)) * 0.5).astype(np.float32), 0.05),           # ~6 points per voxel
    'vox_negative': ((rng.standard_normal((5000, 3)) * 1.5).astype(np.float32), 0.2),  # negative coordinates
    'vox_2cm': ((rng.random((30000, 3)) * np.array([3.0, 2.0, 1.0])).astype(np.float32), 0.02),
    'vox_single': (np.array([[0.3, 0.2, 0.1]], np.float32), 0.1),
}
for name, (coord, vs) in cases.items():
    key = fnv_hash_vec(np.floor(coord / np.array(vs)))
    idx = voxelize(coord, vs, mode='deterministic')
    blobs = dict(coord=coord, voxel=np.float64(vs), key=key, idx=idx.astype(np.int64))
    if name == 'vox_dense':
        for i, part in enumerate(voxelize(coord, vs, mode='multiple')):
            blobs[f'multi{i}'] = part.astype(np.int64)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **blobs)
    print(name, coord.shape[0], 'points ->', idx.shape[0], 'voxels')
