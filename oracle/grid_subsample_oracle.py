"""TEST INFRASTRUCTURE -- CPU restatement of the reference's barycentre grid subsampling; only tests/, smoke() and
bench.py's cpu_baseline leg may import this.

Follows cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:9-110 (+ SampledData,
grid_subsampling.h:14-84; min_point/max_point, cpp_utils/cloud/cloud.cpp:27-67) step by step in float32:

    origin   = floor(min_corner * (1 / dl)) * dl                       (:29-31, float arithmetic)
    nX, nY   = floor((max_corner - origin) / dl) + 1                   (:34-35)
    (i,j,k)  = floor((p - origin) / dl);  key = i + nX*j + nX*nY*k     (:58-61)
    per key: count, float sums of the points / features IN POINT ORDER (:64-72, grid_subsampling.h:45-81)
    out      = sum * float(1.0 / count) for the point (:91), sum / float(count) for the features (:94-98),
               the most frequent label per label column (:103-105)

The reference emits voxels in std::unordered_map order (unspecified); this restatement -- and the HIP kernel -- emit
them in ascending key order.  Ties of the label vote are also unspecified upstream (first maximum in hash-map
order); here the smallest label wins.  Pinned against the reference's own C++ compiled in the build container
(oracle/_ref/libgridsub_ref.so, oracle/Makefile) through tests/golden/gridsub_*.npz: bit-exact points and features
on every fixture (label ties are excluded from the fixtures' comparison).

`subsample` restates datasetCommon.py:384-421 (levels by repeated subsampling of the previous level, normals as the
averaged feature, and the "<= 16 points: keep the previous level" rule).
"""
import numpy as np

F32 = np.float32


def voxel_keys(points, sampleDl):
    """-> (key int64 [N], origin f32 [3], nX, nY) exactly as :24-61 computes them."""
    pts = np.ascontiguousarray(points, F32)
    dl = F32(sampleDl)
    mn, mx = pts.min(0), pts.max(0)
    inv = F32(1) / dl
    origin = (np.floor(mn * inv) * dl).astype(F32)
    nx = int(np.floor((mx[0] - origin[0]) / dl)) + 1
    ny = int(np.floor((mx[1] - origin[1]) / dl)) + 1
    ijk = np.floor((pts - origin) / dl).astype(np.int64)
    return ijk[:, 0] + nx * ijk[:, 1] + nx * ny * ijk[:, 2], origin, nx, ny


def grid_subsampling(points, features=None, labels=None, sampleDl=0.1):
    """-> (sub_points [M,3] f32, sub_features [M,F] f32 or None, sub_labels [M,L] i32 or None), ascending voxel key."""
    pts = np.ascontiguousarray(points, F32)
    n = pts.shape[0]
    if n == 0:
        return (np.zeros((0, 3), F32), None if features is None else np.zeros((0, features.shape[1]), F32),
                None if labels is None else np.zeros((0, np.atleast_2d(labels.T).T.shape[1]), np.int32))
    key, _, _, _ = voxel_keys(pts, sampleDl)
    uniq, vid, count = np.unique(key, return_inverse=True, return_counts=True)
    m = uniq.shape[0]
    sums = np.zeros((m, 3), F32)
    np.add.at(sums, vid, pts)                                  # unbuffered: float32 adds in point order
    recip = (1.0 / count.astype(np.float64)).astype(F32)
    sub_points = sums * recip[:, None]
    sub_features = None
    if features is not None:
        f = np.ascontiguousarray(features, F32)
        fs = np.zeros((m, f.shape[1]), F32)
        np.add.at(fs, vid, f)
        sub_features = fs / count.astype(F32)[:, None]
    sub_labels = None
    if labels is not None:
        lab = np.ascontiguousarray(labels, np.int32).reshape(n, -1)
        sub_labels = np.zeros((m, lab.shape[1]), np.int32)
        order = np.argsort(vid, kind='stable')
        starts = np.concatenate([[0], np.cumsum(count)])
        for v in range(m):
            rows = lab[order[starts[v]:starts[v + 1]]]
            for c in range(lab.shape[1]):
                vals, cnt = np.unique(rows[:, c], return_counts=True)
                sub_labels[v, c] = vals[np.argmax(cnt)]        # ties: smallest label
    return sub_points, sub_features, sub_labels


def label_vote_is_unique(points, labels, sampleDl):
    """[M, L] bool: True where a voxel's most frequent label is unique (the reference is unspecified elsewhere)."""
    key, _, _, _ = voxel_keys(points, sampleDl)
    uniq, vid, count = np.unique(key, return_inverse=True, return_counts=True)
    lab = np.ascontiguousarray(labels, np.int32).reshape(len(key), -1)
    ok = np.ones((len(uniq), lab.shape[1]), bool)
    order = np.argsort(vid, kind='stable')
    starts = np.concatenate([[0], np.cumsum(count)])
    for v in range(len(uniq)):
        rows = lab[order[starts[v]:starts[v + 1]]]
        for c in range(lab.shape[1]):
            _, cnt = np.unique(rows[:, c], return_counts=True)
            ok[v, c] = (cnt == cnt.max()).sum() == 1
    return ok


def subsample(coord, norm, grid_size):
    """datasetCommon.py:384-421: level 0 = the input; level j = barycentre subsampling of level j-1 at grid_size[j]
    with the normals averaged; a level of <= 16 points repeats the previous one."""
    point_list, norm_list = [], []
    for j, gs in enumerate(grid_size):
        if j == 0:
            p, f = np.asarray(coord, F32), np.asarray(norm, F32)
        else:
            p, f, _ = grid_subsampling(point_list[-1], norm_list[-1], None, gs)
            if p.shape[0] <= 16:
                p, f = point_list[-1], norm_list[-1]
        point_list.append(p)
        norm_list.append(f)
    return point_list, norm_list


def lex_order(points):
    """Canonical row order for comparing against the reference's hash-map order: lexicographic in (x, y, z)."""
    return np.lexsort((points[:, 2], points[:, 1], points[:, 0]))
