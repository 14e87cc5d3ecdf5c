"""GPU: the drop-in `knn_post_dataloader_utils` (compute_knn_packed / prepare / compute_knn) against the
oracle's per-sample brute force, bit-exact, on a 2-sample, 3-level packed batch as the training loop
builds it (train_ScanNet_DDP_WarmUP.py:382-383)."""
import numpy as np
import pytest
import torch

from oracle import pcf_oracle as O

pytestmark = pytest.mark.gpu


def _batch(seed=0):
    rng = np.random.default_rng(seed)
    counts = [[900, 500], [300, 170], [90, 60]]            # [level][sample]
    clouds = [[rng.random((n, 3), dtype=np.float32) * 2 for n in lvl] for lvl in counts]
    pointclouds = [torch.from_numpy(np.concatenate(lvl))[None] for lvl in clouds]
    return counts, clouds, pointclouds


def test_compute_knn_packed_and_prepare(device):
    import knn_post_dataloader_utils as U
    counts, clouds, pointclouds = _batch()
    K_self, K_fwd, K_prop = [16, 16, 8], [None, 16, 8], [None, 16, 4]
    es, ef, ep = U.compute_knn_packed(pointclouds, counts, K_self, K_fwd, K_prop)
    S, L = 2, 3
    assert len(es) == S and len(es[0]) == L and len(ef[0]) == L - 1 and len(ep[0]) == L - 1
    # per-sample local indices, as the reference returns them
    for s in range(S):
        for j in range(L):
            np.testing.assert_array_equal(es[s][j].cpu().numpy(), O.knn_bruteforce(clouds[j][s], clouds[j][s], K_self[j]))
        for j in range(1, L):
            np.testing.assert_array_equal(ef[s][j - 1].cpu().numpy(), O.knn_bruteforce(clouds[j - 1][s], clouds[j][s], K_fwd[j]))
            np.testing.assert_array_equal(ep[s][j - 1].cpu().numpy(), O.knn_bruteforce(clouds[j][s], clouds[j - 1][s], K_prop[j]))
    ps, pf, pp = U.prepare(es, ef, ep)
    off = [np.concatenate([[0], np.cumsum(c)]) for c in counts]
    for j in range(L):
        want = O.knn_packed(np.concatenate(clouds[j]), np.concatenate(clouds[j]), off[j], off[j], K_self[j])
        assert ps[j].shape == (1, sum(counts[j]), K_self[j]) and ps[j].dtype == torch.int64
        np.testing.assert_array_equal(ps[j][0].cpu().numpy(), want)
    for j in range(1, L):
        want = O.knn_packed(np.concatenate(clouds[j - 1]), np.concatenate(clouds[j]), off[j - 1], off[j], K_fwd[j])
        np.testing.assert_array_equal(pf[j - 1][0].cpu().numpy(), want)
        want = O.knn_packed(np.concatenate(clouds[j]), np.concatenate(clouds[j - 1]), off[j], off[j - 1], K_prop[j])
        np.testing.assert_array_equal(pp[j - 1][0].cpu().numpy(), want)
    # the generic path (lists of numpy arrays with a -1 padding entry, as the CPU dataloader emits them)
    as_np = lambda nested: [[t.cpu().numpy() for t in lvl] for lvl in nested]
    es_n, ef_n, ep_n = as_np(es), as_np(ef), as_np(ep)
    es_n[1][0][0, 3] = -1
    gs, gf, gp = U.prepare(es_n, ef_n, ep_n)
    ref = ps[0][0].cpu().clone()
    ref[counts[0][0], 3] = -1
    assert torch.equal(gs[0][0], ref)
    for a, b in zip(gf + gp, pf + pp):
        assert torch.equal(a.cpu(), b.cpu())


def test_compute_knn_and_inverse(device):
    import knn_post_dataloader_utils as U
    rng = np.random.default_rng(3)
    ref = rng.random((400, 3), dtype=np.float32)
    qry = rng.random((150, 3), dtype=np.float32)
    got = U.compute_knn(ref, torch.from_numpy(qry), 8)
    np.testing.assert_array_equal(got.cpu().numpy(), O.knn_bruteforce(ref, qry, 8))
    got = U.compute_knn(ref, qry, 4, dilated_rate=3)
    np.testing.assert_array_equal(got.cpu().numpy(), O.knn_bruteforce(ref, qry, 12)[:, ::3])
    few = U.compute_knn(ref[:5], qry, 8)                       # fewer refs than K: random valid indices
    assert few.shape == (150, 8) and int(few.min()) >= 0 and int(few.max()) < 5
    # CSR transposes of every edge set, the structure util/common_util.py:250-327 returns
    counts, clouds, pointclouds = _batch(1)
    edges = U.prepare(*U.compute_knn_packed(pointclouds, counts, [8, 8, 8], [None, 8, 8], [None, 8, 8]))
    pcs = [p.to(device) for p in pointclouds]
    inv_self, inv_fwd, inv_prop = U.compute_knn_inverse(pcs, *edges)
    for inv, es, lvls in ((inv_self, edges[0], [0, 1, 2]), (inv_fwd, edges[1], [0, 1]), (inv_prop, edges[2], [0, 1])):
        assert len(inv) == 3 and all(len(x) == len(es) for x in inv)
        for j, e in enumerate(es):
            total = pointclouds[lvls[j]].shape[1]
            wn, wk, wi = O.knn_inverse(e[0].cpu().numpy(), total)
            np.testing.assert_array_equal(inv[0][j][0].cpu().numpy(), wn)
            np.testing.assert_array_equal(inv[1][j][0].cpu().numpy(), wk)
            np.testing.assert_array_equal(inv[2][j][0].cpu().numpy(), wi)
