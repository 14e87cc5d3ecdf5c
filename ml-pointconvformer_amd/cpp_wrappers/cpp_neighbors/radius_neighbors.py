"""Drop-in for the reference's nanoflann extension ``cpp_wrappers.cpp_neighbors.radius_neighbors`` (built from
cpp_wrappers/cpp_neighbors/wrapper.cpp; imported at datasetCommon.py:10-13 and called at :512-524) on the HIP kNN.

The reference module cannot link as shipped; its one entry point used by the code base is

    batch_kquery(queries, supports, q_batches, s_batches, K) -> uint64 [Nq, K]     (wrapper.cpp:293-429)

K nearest supports of every query, batch element by batch element, as indices into the PACKED support array (local index
+ the offset of the batch element, neighbors.cpp:318-327), ascending by distance; slots a batch element cannot fill (fewer
than K supports) hold ``len(supports)``.  Here: one exact kNN launch set over the packed batch (pcf_cuda.knn_packed,
squared L2 in difference form, ties -> lower index), result copied back to the host as the reference returns numpy.
``cpp_wrappers`` and ``cpp_wrappers.cpp_neighbors`` are namespace packages (no ``__init__``) on both sides, so with
``ml-pointconvformer_amd/`` first on ``PYTHONPATH`` the reference's import statement finds this module.
"""
import numpy as np
import torch

import pcf_cuda


def _offsets(counts, dev):
    off = np.zeros(len(counts) + 1, np.int32)
    np.cumsum(np.asarray(counts, np.int64), out=off[1:])
    return off, torch.from_numpy(off).to(dev)


def batch_kquery(queries, supports, q_batches, s_batches, K=16):
    """queries [Nq,3], supports [Ns,3] float32 (numpy or tensors); q_batches / s_batches: per-element lengths."""
    if not torch.cuda.is_available():
        raise RuntimeError('cpp_neighbors.batch_kquery: no GPU visible; the HIP kNN has no CPU fallback')
    dev = torch.device('cuda', torch.cuda.current_device())
    q = torch.as_tensor(np.asarray(queries, np.float32) if not torch.is_tensor(queries) else queries, dtype=torch.float32).reshape(-1, 3)
    s = torch.as_tensor(np.asarray(supports, np.float32) if not torch.is_tensor(supports) else supports, dtype=torch.float32).reshape(-1, 3)
    qb = np.atleast_1d(np.asarray(q_batches)).astype(np.int64)
    sb = np.atleast_1d(np.asarray(s_batches)).astype(np.int64)
    if qb.sum() != q.shape[0] or sb.sum() != s.shape[0] or len(qb) != len(sb):
        raise RuntimeError('batch_kquery: batch lengths must sum to the number of queries / supports')
    _, qoff = _offsets(qb, dev)
    _, soff = _offsets(sb, dev)
    idx = pcf_cuda.knn_packed(s.to(dev).contiguous(), q.to(dev).contiguous(), soff, qoff, int(K))      # packed support indices, -1 = none
    idx = torch.where(idx < 0, torch.full_like(idx, s.shape[0]), idx)
    return idx.cpu().numpy().astype(np.uint64)
