"""Generates tests/golden/gridsub_*.npz: seeded inputs and what THE REFERENCE'S OWN C++ grid subsampling returns for
them.  Run in the build container only (needs oracle/_ref/libgridsub_ref.so = the reference's
cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp compiled by `make -C oracle ref`):

    python tests/golden/make_gridsub_golden.py

Each fixture holds: points [N,3] f32, features [N,F] f32 (F may be 0), labels [N,1] i32, sampleDl, and the
reference's outputs ref_points / ref_features / ref_labels with the rows put in lexicographic (x, y, z) order of
ref_points (the reference emits std::unordered_map order).  `ref_label_unique` marks the voxels whose label vote
has a unique winner (ties are unspecified upstream).  The multi-level fixture chains the reference over the YAML's
grid sizes exactly as datasetCommon.subsample does (:384-421).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import grid_subsample_oracle as G      # noqa: E402  (canonical order + unique-vote mask only)
from oracle import gridsub_ref as R                # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sheet(rng, n, side):
    xy = rng.random((n, 2)) * side
    z = 0.35 * np.sin(1.1 * xy[:, 0]) + 0.25 * np.cos(0.7 * xy[:, 1])
    p = np.stack([xy[:, 0], xy[:, 1], z], 1).astype(np.float32)
    nrm = np.stack([-0.385 * np.cos(1.1 * xy[:, 0]), 0.175 * np.sin(0.7 * xy[:, 1]), np.ones(n)], 1)
    return p, (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)


def cases():
    rng = np.random.default_rng(20240607)
    p = (rng.random((3000, 3)) * 2.0).astype(np.float32)
    yield 'volume', p, rng.standard_normal((3000, 3)).astype(np.float32), 0.1
    p, f = sheet(rng, 4000, 6.0)
    yield 'surface', p, f, 0.08
    p = (rng.random((2500, 3)) * 1.0 + 0.01).astype(np.float32)          # ~40 points per voxel: summation order matters
    yield 'dense', p, rng.standard_normal((2500, 5)).astype(np.float32), 0.25
    p = (rng.random((1500, 3)) * 3.0 - 7.3).astype(np.float32)           # negative coordinates, no features
    yield 'negative', p, None, 0.2
    p = np.repeat((rng.random((40, 3)) * 1.5).astype(np.float32), 9, 0)  # exact duplicates
    yield 'duplicates', p, rng.standard_normal((360, 3)).astype(np.float32), 0.05
    yield 'single', np.array([[0.3, -1.2, 4.0]], np.float32), np.array([[1.0, 2.0, 3.0]], np.float32), 0.1
    p = (np.round(rng.random((2000, 3)) * 40) * 0.05).astype(np.float32)  # points on voxel faces: floor() of exact multiples
    yield 'lattice', p, rng.standard_normal((2000, 3)).astype(np.float32), 0.05


def main():
    assert R.available(), 'build oracle/_ref first: make -C oracle ref'
    rng = np.random.default_rng(7)
    for name, p, f, dl in cases():
        lab = rng.integers(0, 20, (p.shape[0], 1)).astype(np.int32)
        rp, rf, rl = R.grid_subsampling(p, f, lab, dl)
        o = G.lex_order(rp)
        # the unique-vote mask comes out in ascending-key order: map it through the oracle's own row order
        op, _, _ = G.grid_subsampling(p, f, lab, dl)
        uniq = G.label_vote_is_unique(p, lab, dl)[G.lex_order(op)]
        blobs = dict(points=p, features=f if f is not None else np.zeros((p.shape[0], 0), np.float32), labels=lab,
                     sampleDl=np.float32(dl), ref_points=rp[o], ref_labels=rl[o], ref_label_unique=uniq,
                     ref_features=rf[o] if rf is not None else np.zeros((rp.shape[0], 0), np.float32))
        np.savez_compressed(os.path.join(HERE, f'gridsub_{name}.npz'), **blobs)
        print(name, p.shape[0], '->', rp.shape[0], 'voxels')
    # multi-level chain (configPCF_10cm_lite grid sizes) of one 6000-point sheet through the reference
    p, f = sheet(np.random.default_rng(11), 6000, 8.0)
    grid = [0.1, 0.2, 0.4, 0.8, 1.6]
    blobs = dict(points=p, features=f, grid_size=np.asarray(grid, np.float32))
    lp, lf = p, f
    for j, gs in enumerate(grid[1:], 1):
        rp, rf, _ = R.grid_subsampling(lp, lf, None, gs)
        # the next level's input order matters for its float sums: keep the canonical (ascending voxel key) order that
        # the oracle and the HIP kernel emit -- ascending key order == the oracle's output order
        op, of, _ = G.grid_subsampling(lp, lf, None, gs)
        assert np.array_equal(op[G.lex_order(op)], rp[G.lex_order(rp)]) and np.array_equal(of[G.lex_order(op)], rf[G.lex_order(rp)])
        if op.shape[0] <= 16:
            op, of = lp, lf
        blobs[f'level{j}_points'], blobs[f'level{j}_features'] = op, of
        lp, lf = op, of
        print('level', j, op.shape[0])
    np.savez_compressed(os.path.join(HERE, 'gridsub_levels.npz'), **blobs)


if __name__ == '__main__':
    main()
