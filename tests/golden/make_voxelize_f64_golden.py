"""Golden vectors for the level-0 voxelisation of FLOAT64 coordinates, from the reference's own function
(util/voxelize.py:44-82).  Run in the build container:  python tests/golden/make_voxelize_f64_golden.py

A loader that keeps its coordinates in double hands np.floor(coord / voxel_size) double quotients under any NumPy
version; the points of this fixture sit within 1e-9 .. 1e-12 of voxel faces (plus a uniform cloud), where hashing the
float32 roundings instead would put many of them into the neighbouring voxel.  Stored: coordinates (float64), voxel size,
the reference's FNV keys per point, its 'deterministic' selection."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
from util.voxelize import fnv_hash_vec, voxelize      # noqa: E402

rng = np.random.default_rng(11)
vs = 0.02
faces = rng.integers(-200, 200, (6000, 3)).astype(np.float64) * vs
faces += rng.choice([-1.0, 1.0], faces.shape) * 10.0 ** rng.uniform(-12, -9, faces.shape)          # just off a face, both sides
cloud = rng.random((6000, 3)) * np.array([3.0, 2.0, 1.0]) - 0.5
coord = np.concatenate([faces, cloud]).astype(np.float64)
key = fnv_hash_vec(np.floor(coord / np.array(vs)))
idx = voxelize(coord, vs, mode='deterministic')
as32 = fnv_hash_vec(np.floor(coord.astype(np.float32).astype(np.float64) / np.array(vs)))
print('points whose key changes when the coordinates are rounded to float32 first:', int((as32 != key).sum()), 'of', coord.shape[0])
np.savez_compressed(os.path.join(HERE, 'vox_f64_faces.npz'), coord=coord, voxel=np.float64(vs), key=key, idx=idx.astype(np.int64))
print('vox_f64_faces', coord.shape[0], 'points ->', idx.shape[0], 'voxels')
