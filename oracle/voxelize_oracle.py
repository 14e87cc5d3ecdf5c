"""CPU oracle of the level-0 voxelisation  --  TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

Restates util/voxelize.py:10-22 (FNV64-1A over the floored quotient) and :44-82 (one point per voxel) of the reference in
numpy.  Parity status: PINNED -- tests/test_voxelize_cpu.py holds the keys bit-exactly and the selected voxel sequence
to fixtures produced by importing the reference's own function (tests/golden/make_voxelize_golden.py).  The reference
picks "the first point after an UNSTABLE argsort" in its deterministic mode; this oracle (and the HIP kernel) pick the
lowest point index of the voxel, so selections are compared voxel by voxel (same voxel sequence), and the choice inside
a voxel by its defining property."""
import numpy as np

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)


def fnv_keys(coord, voxel_size):
    """uint64 [N]: hash of floor(coord / voxel_size) per axis, the quotient in float64 (numpy promotes the float32
    coordinates against the 0-d float64 array np.array(voxel_size), util/voxelize.py:58)."""
    coord = np.asarray(coord)
    if coord.dtype != np.float64:                          # float64 input: the quotient of the doubles themselves (any numpy)
        coord = coord.astype(np.float32).astype(np.float64)
    d = np.floor(coord / np.float64(voxel_size))
    a = d.astype(np.int64).astype(np.uint64)              # two's complement of negative whole numbers
    h = np.full(a.shape[0], FNV_OFFSET, np.uint64)
    with np.errstate(over='ignore'):
        for j in range(a.shape[1]):
            h = h * FNV_PRIME
            h = np.bitwise_xor(h, a[:, j])
    return h


def voxelize(coord, voxel_size, mode='deterministic', rank=0):
    """-> (idx int64 [V] in ascending key order, counts [V]).  deterministic: lowest index of the voxel; rank: point
    `rank % count` of the voxel in index order."""
    key = fnv_keys(coord, voxel_size)
    order = np.argsort(key, kind='stable')
    ks = key[order]
    head = np.ones(ks.shape[0], bool)
    head[1:] = ks[1:] != ks[:-1]
    start = np.flatnonzero(head)
    counts = np.diff(np.append(start, ks.shape[0]))
    off = 0 if mode == 'deterministic' else rank % counts
    return order[start + off].astype(np.int64), counts
