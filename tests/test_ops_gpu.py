"""GPU: the HIP operators (through the C ABI, via the drop-in `pcf_cuda` module) against the oracle.

Bar (BASELINE.json north_star): indices bit-exact, features / gradients within 1e-3 in fp32.  The
comparisons here use rtol = atol = 2e-4 -- tighter than the bar; what differs is summation order.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, split
from oracle import pcf_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=2e-4, atol=2e-4)


def _mk(shape, gen, dev, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(dev)


def _case(B, N, Nout, K, Ci, Ca, Cm, H, seed, bad_frac=0.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, Ci, generator=g)
    idx = torch.randint(0, max(N, 1), (B, Nout, K), generator=g)
    if bad_frac > 0:
        bad = torch.rand(B, Nout, K, generator=g) < bad_frac
        idx = torch.where(bad, torch.where(torch.rand(B, Nout, K, generator=g) < 0.5, -1, N + 3), idx)
    guid = torch.rand(B, Nout, K, max(H, 1), generator=g)
    w = torch.randn(B, Nout, K, Cm, generator=g)
    add = torch.randn(B, Nout, K, Ca, generator=g)
    return x, idx, guid, w, add


def _safe(idx, N):
    """oracle-side view of out-of-range neighbours: they contribute nothing."""
    ok = (idx >= 0) & (idx < N)
    return idx.clamp(0, max(N - 1, 0)), ok


PCF_SHAPES = [
    # B, N, Nout, K, Ci, Cm, H
    (1, 500, 500, 16, 16, 16, 8),      # BASELINE layer shape (C=64 -> Ci=16)
    (1, 300, 100, 8, 32, 4, 4),        # strided, Cm=4
    (2, 64, 40, 5, 12, 3, 3),          # batch 2, odd K, Cm and H not powers of two (generic path)
    (1, 200, 200, 16, 96, 16, 8),      # widest PCFLayer of configPCF_10cm
    (1, 130, 130, 16, 48, 8, 8),
    (1, 90, 90, 16, 64, 32, 1),
    (1, 33, 77, 3, 7, 1, 7),           # Cm=1, odd Ci
    (1, 400, 150, 16, 32, 4, 8),       # 10cm-lite PCFLayers (C_mid = 4): matrix-core kernels, strided
    (2, 90, 90, 16, 64, 4, 8),         # ... batch 2, four channel tiles
    (1, 150, 150, 16, 48, 4, 8),       # ... three channel tiles
    (2, 120, 70, 16, 32, 16, 8),       # C_mid = 16, two channel tiles, batch 2, strided
]


ENGINES = ['default', 'lds', 'tiled']


@pytest.fixture
def engine(request):
    """Run a test under one of the aggregate kernel families (pcf_hip_set_aggregate_engine): the default dispatch, the
    LDS-tiled kernels everywhere (the cross-check of the matrix-core ones), the tiled matrix-core kernels everywhere."""
    import pcf_cuda
    pcf_cuda.set_aggregate_engine(request.param)
    assert pcf_cuda.get_aggregate_engine() == request.param
    yield request.param
    pcf_cuda.set_aggregate_engine('default')


@pytest.mark.parametrize('engine', ENGINES, indirect=True)
@pytest.mark.parametrize('shape', PCF_SHAPES)
def test_pcf_forward_backward(device, shape, engine):
    import pcf_cuda
    B, N, Nout, K, Ci, Cm, H = shape
    x, idx, guid, w, _ = _case(B, N, Nout, K, Ci, 0, Cm, H, seed=sum(shape))
    out = pcf_cuda.pcf_forward(x.to(device), idx.to(device), guid.to(device), w.to(device))
    want = O.pcf_forward(x, idx, guid, w)
    torch.testing.assert_close(out.cpu(), want, **TOL)
    gout = torch.randn(want.shape, generator=torch.Generator().manual_seed(7))
    gx, gg, gw = pcf_cuda.pcf_backward(gout.to(device), x.to(device), idx.to(device), guid.to(device), w.to(device))
    wx, wg, ww = O.pcf_backward(gout, x, idx, guid, w)
    torch.testing.assert_close(gw.cpu(), ww, **TOL)
    torch.testing.assert_close(gg.cpu(), wg, **TOL)
    torch.testing.assert_close(gx.cpu(), wx, **TOL)


PCONV_SHAPES = [
    # B, N, Nout, K, Ci, Ca, Cm
    (1, 256, 256, 16, 3, 0, 16),       # test_configs/pointconv_single.yaml layer
    (1, 144, 144, 16, 6, 12, 16),      # level-0 PointConv of configPCF_10cm
    (1, 240, 90, 12, 16, 16, 16),      # PointConvStridePE
    (1, 60, 200, 16, 128, 16, 1),      # PointConvTransposePE: N < Nout, Cm=1
    (1, 50, 120, 16, 384, 32, 1),      # widest decoder layer
    (2, 70, 70, 4, 8, 4, 4),
    (1, 100, 100, 16, 16, 5, 16),      # Ca not a multiple of 4 -> scalar rows
    (1, 40, 40, 64, 16, 16, 16),       # K=64, the shape of test_kernels.py:1090-1094
    (1, 300, 120, 16, 16, 16, 16),     # StridePE of the BASELINE configs, K = 16: unguided matrix-core kernels
    (2, 90, 90, 16, 16, 16, 4),        # ... 10cm-lite (C_mid = 4), batch 2
    (1, 200, 200, 16, 32, 16, 4),      # three channel tiles
    (1, 150, 150, 16, 32, 0, 16),      # no appended features
    (1, 40, 110, 16, 384, 32, 3),      # configPCF_2cm_PTF2 decoder (mid_dim_back 3), widest layer
    (1, 90, 250, 16, 128, 16, 3),      # ... narrowest
]


@pytest.mark.parametrize('engine', ENGINES, indirect=True)
@pytest.mark.parametrize('shape', PCONV_SHAPES)
def test_pconv_forward_backward(device, shape, engine):
    import pcf_cuda
    B, N, Nout, K, Ci, Ca, Cm = shape
    x, idx, _, w, add = _case(B, N, Nout, K, Ci, Ca, Cm, 1, seed=sum(shape))
    d = lambda t: t.to(device)
    out = pcf_cuda.pconv_forward(d(x), d(idx), d(w), d(add))
    want = O.pconv_forward(x, idx, w, add)
    torch.testing.assert_close(out.cpu(), want, **TOL)
    gout = torch.randn(want.shape, generator=torch.Generator().manual_seed(11))
    gx, gw, ga = pcf_cuda.pconv_backward(d(gout), d(x), d(idx), d(w), d(add))
    wx, ww, wa = O.pconv_backward(gout, x, idx, w, add)
    torch.testing.assert_close(gw.cpu(), ww, **TOL)
    torch.testing.assert_close(ga.cpu(), wa, **TOL)
    torch.testing.assert_close(gx.cpu(), wx, **TOL)


def test_aggregate_engines_select_their_kernels(device):
    """The engine switch really changes the kernels (launch log), and the three families agree with each other to fp32
    rounding at the BASELINE shape (Ci = Cm = 16, H = 8, K = 16) and at C_mid = 4."""
    import pcf_cuda
    want = {('default', 16): ('agg_fwd_fx_mfma_kernel', 'agg_bwd_fx_mfma_kernel'),
            ('lds', 16): ('agg_fwd_kernel<16,true,true>', 'agg_bwd_kernel<16,true,fx>'),
            ('tiled', 16): ('agg_fwd_mfma_kernel<16,0>', 'agg_bwd_mfma_kernel<16,0>'),
            ('default', 4): ('agg_fwd_mfma_kernel<4,0>', 'agg_bwd_mfma_kernel<4,0>'),
            ('lds', 4): ('agg_fwd_kernel<4,true>', 'agg_bwd_kernel<4,true>')}
    d = lambda t: t.to(device)
    try:
        for Cm in (16, 4):
            x, idx, guid, w, _ = _case(1, 700, 700, 16, 16, 0, Cm, 8, seed=31 + Cm)
            gout = torch.randn(1, 700, 16 * Cm, generator=torch.Generator().manual_seed(2))
            res = {}
            for eng in ENGINES:
                if (eng, Cm) not in want:
                    continue
                pcf_cuda.set_aggregate_engine(eng)
                pcf_cuda.launch_log(True)
                out = pcf_cuda.pcf_forward(d(x), d(idx), d(guid), d(w))
                grads = pcf_cuda.pcf_backward(d(gout), d(x), d(idx), d(guid), d(w))
                log = pcf_cuda.read_launch_log()
                assert log == list(want[(eng, Cm)]), (eng, Cm, log)
                res[eng] = [out.cpu()] + [t.cpu() for t in grads]
            for eng in res:
                for a, b in zip(res[eng], res['default']):
                    torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-5 * float(b.abs().max()))
    finally:
        pcf_cuda.launch_log(False)
        pcf_cuda.set_aggregate_engine('default')


@pytest.mark.parametrize('Cm', [16, 1])
def test_out_of_range_neighbours_are_ignored(device, Cm):
    import pcf_cuda
    B, N, Nout, K, Ci, Ca, H = 1, 120, 80, 16, 16, 4, 8
    x, idx, guid, w, add = _case(B, N, Nout, K, Ci, Ca, Cm, H, seed=5, bad_frac=0.2)
    sidx, ok = _safe(idx, N)
    d = lambda t: t.to(device)
    out = pcf_cuda.pcf_forward(d(x), d(idx), d(guid), d(w))
    torch.testing.assert_close(out.cpu(), O.pcf_forward(x, sidx, guid * ok[..., None], w), **TOL)
    out = pcf_cuda.pconv_forward(d(x), d(idx), d(w), d(add))
    xz = torch.cat([x, torch.zeros(B, 1, Ci)], 1)                      # row N = zeros
    want = O.pconv_forward(xz, torch.where(ok, idx, N), w, add)
    torch.testing.assert_close(out.cpu(), want, **TOL)
    gout = torch.randn(want.shape, generator=torch.Generator().manual_seed(3))
    gx, gw, ga = pcf_cuda.pconv_backward(d(gout), d(x), d(idx), d(w), d(add))
    wx, ww, wa = O.pconv_backward(gout, xz, torch.where(ok, idx, N), w, add)
    torch.testing.assert_close(gx.cpu(), wx[:, :N], **TOL)
    torch.testing.assert_close(gw.cpu(), ww, **TOL)
    torch.testing.assert_close(ga.cpu(), wa, **TOL)


def test_empty_inputs(device):
    import pcf_cuda
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)
    out = pcf_cuda.pcf_forward(z(1, 10, 8), z(1, 0, 4, dt=torch.long), z(1, 0, 4, 2), z(1, 0, 4, 4))
    assert out.shape == (1, 0, 32)
    gx, gg, gw = pcf_cuda.pcf_backward(z(1, 0, 32), z(1, 10, 8), z(1, 0, 4, dt=torch.long), z(1, 0, 4, 2), z(1, 0, 4, 4))
    assert gx.shape == (1, 10, 8) and float(gx.abs().sum()) == 0.0
    inv_n, inv_k, inv_idx = pcf_cuda.compute_knn_inverse(z(1, 0, 4, dt=torch.long), 5)
    assert inv_idx.cpu().tolist() == [[0] * 6] and inv_n.shape == (1, 0)


@pytest.mark.parametrize('name,has_guid', [('pcf_self_64', True), ('pcf_self_32_64', True), ('pcf_strided', True)])
def test_pcf_against_reference_golden(device, name, has_guid):
    """Tensors the reference's own PCFLayer produced around its aggregate (tests/golden)."""
    import pcf_cuda
    g = load_golden(name)
    d = lambda t: t.contiguous().to(device)
    out = pcf_cuda.pcf_forward(d(g['cap.fx']), d(g['in.nei_inds']), d(g['cap.score']), d(g['cap.w']))
    torch.testing.assert_close(out.cpu(), g['cap.agg'], **TOL)
    gx, gg, gw = pcf_cuda.pcf_backward(d(g['gcap.agg']), d(g['cap.fx']), d(g['in.nei_inds']), d(g['cap.score']),
                                       d(g['cap.w']))
    torch.testing.assert_close(gg.cpu(), g['gcap.score'], **TOL)
    torch.testing.assert_close(gw.cpu(), g['gcap.w'], **TOL)


def test_pconv_linear_against_reference_golden(device):
    import pcf_cuda
    d = lambda t: t.contiguous().to(device)
    # Ca = 0 (pointconv_single): forward, backward (atomics) and opt backward (CSR)
    g = load_golden('pointconv_single')
    x, idx, w = g['in.dense_feats'], g['in.nei_inds'], g['cap.w']
    add = torch.zeros(1, idx.shape[1], idx.shape[2], 0)
    lw, lb = g['sd.linear.weight'], g['sd.linear.bias']
    for fwd in (pcf_cuda.pconv_linear_forward, pcf_cuda.pconv_linear_cutlass_forward):
        out, p = fwd(d(x), d(idx), d(w), d(add), d(lw), d(lb))
        torch.testing.assert_close(p.cpu(), g['cap.agg'], **TOL)
        torch.testing.assert_close(out.cpu(), g['cap.lin'], **TOL)
    want = (g['gin.dense_feats'], g['gcap.w'], None, g['gsd.linear.weight'], g['gsd.linear.bias'])
    got = pcf_cuda.pconv_linear_backward(d(g['gcap.lin']), d(x), d(idx), d(w), d(add), d(lw), p)
    inv = pcf_cuda.compute_knn_inverse(d(idx), x.shape[1])
    got2 = pcf_cuda.pconv_linear_opt_backward(d(g['gcap.lin']), d(x), *inv, d(idx), d(w), d(add), d(lw), p)
    for res in (got, got2):
        for a, b in zip(res, want):
            if b is not None:
                torch.testing.assert_close(a.cpu(), b, **TOL)
    # Ca = 16, strided (stride_pe) and Cm = 1, N < Nout (transpose_pe): aggregate part
    for name, xk in (('stride_pe', 'cap.fx'), ('transpose_pe', 'in.sparse_feats'), ('pointconv_vi_pe', 'in.dense_feats')):
        g = load_golden(name)
        add = g['cap.pe'] if 'cap.pe' in g else g['out.wn_in']
        lw = g['sd.linear.c.weight']
        lb = g['sd.linear.c.bias']
        out, p = pcf_cuda.pconv_linear_forward(d(g[xk]), d(g['in.nei_inds']), d(g['cap.w']), d(add), d(lw), d(lb))
        torch.testing.assert_close(p.cpu(), g['cap.agg'], **TOL)
        torch.testing.assert_close(out.cpu(), g['cap.lin'], **TOL)


LIN_SHAPES = [
    # B, N, Nout, K, Ci, Ca, Cm, Co
    (1, 300, 300, 16, 6, 12, 16, 64),     # 288 -> 64
    (1, 300, 120, 16, 16, 16, 16, 32),    # 512 -> 32 (narrow tile)
    (1, 80, 250, 16, 128, 16, 1, 64),     # 144 -> 64, N < Nout
    (2, 90, 90, 8, 8, 4, 4, 16),
    (1, 1000, 1000, 16, 3, 0, 16, 32),    # enough points for a split-K reduction of grad_lin_w
    (1, 64, 64, 16, 3, 0, 1, 5),          # J = 3: scalar GEMM loads
    (1, 300, 120, 16, 16, 16, 4, 32),     # StridePE of the 10cm-lite model: unguided matrix-core kernels, CSR contribution rows
    (2, 80, 80, 16, 16, 16, 16, 32),      # ... C_mid = 16, batch 2
    (2, 70, 110, 5, 192, 32, 1, 24),      # C_mid = 1 kernels: batch 2, odd K, 48 quads per row (not a power of two)
    (1, 50, 120, 16, 384, 32, 1, 40),     # C_mid = 1, widest decoder layer: two lane passes per row
    (1, 90, 60, 16, 8, 0, 1, 12),         # C_mid = 1, two quads per row: 32 list entries per wave step
    (1, 40, 110, 16, 384, 32, 3, 256),    # configPCF_2cm_PTF2 decoder (mid_dim_back 3): 1248 -> 256
    (1, 90, 250, 16, 128, 16, 3, 64),     # ... 432 -> 64
]


@pytest.mark.parametrize('shape', LIN_SHAPES)
def test_pconv_linear_forward_backward(device, shape):
    import pcf_cuda
    B, N, Nout, K, Ci, Ca, Cm, Co = shape
    x, idx, _, w, add = _case(B, N, Nout, K, Ci, Ca, Cm, 1, seed=sum(shape))
    g = torch.Generator().manual_seed(99)
    J = (Ci + Ca) * Cm
    lw = torch.randn(Co, J, generator=g) / J ** 0.5
    lb = torch.randn(Co, generator=g)
    d = lambda t: t.to(device)
    out, p = pcf_cuda.pconv_linear_forward(d(x), d(idx), d(w), d(add), d(lw), d(lb))
    wout, wp = O.pconv_linear_forward(x, idx, w, add, lw, lb)
    torch.testing.assert_close(p.cpu(), wp, **TOL)
    torch.testing.assert_close(out.cpu(), wout, **TOL)
    gout = torch.randn(wout.shape, generator=g)
    want = O.pconv_linear_backward(gout, x, idx, w, add, lw, wp)
    got = pcf_cuda.pconv_linear_backward(d(gout), d(x), d(idx), d(w), d(add), d(lw), p)
    inv = pcf_cuda.compute_knn_inverse(d(idx), N)
    got2 = pcf_cuda.pconv_linear_opt_backward(d(gout), d(x), *inv, d(idx), d(w), d(add), d(lw), p)
    scale = max(1.0, float(want[3].abs().max()))
    for res in (got, got2):
        for i, (a, b) in enumerate(zip(res, want)):
            tol = dict(rtol=2e-4, atol=2e-4 * scale) if i >= 3 else TOL
            torch.testing.assert_close(a.cpu(), b, **tol)
    # the CSR path is deterministic: bitwise identical across runs
    again = pcf_cuda.pconv_linear_opt_backward(d(gout), d(x), *inv, d(idx), d(w), d(add), d(lw), p)
    assert torch.equal(again[0], got2[0])


def test_opt_backward_rejects_short_csr(device):
    import pcf_cuda
    B, N, Nout, K, Ci, Ca, Cm, Co = 1, 50, 50, 4, 4, 0, 4, 8
    x, idx, _, w, add = _case(B, N, Nout, K, Ci, Ca, Cm, 1, seed=1)
    d = lambda t: t.to(device)
    lw = torch.randn(Co, 16)
    out, p = pcf_cuda.pconv_linear_forward(d(x), d(idx), d(w), d(add), d(lw), d(torch.zeros(Co)))
    inv_n, inv_k, inv_idx = pcf_cuda.compute_knn_inverse(d(idx), N)
    with pytest.raises(RuntimeError, match='inverse_neighbor_idx size must be'):
        pcf_cuda.pconv_linear_opt_backward(d(torch.zeros(1, Nout, Co)), d(x), inv_n, inv_k, inv_idx[:, :N].contiguous(),
                                           d(idx), d(w), d(add), d(lw), p)
    with pytest.raises(RuntimeError, match='must be contiguous'):
        pcf_cuda.pconv_forward(d(x).transpose(1, 2).transpose(1, 2)[:, ::2], d(idx), d(w), d(add))


@pytest.mark.parametrize('Nq,K,total,seed', [(500, 16, 500, 0), (300, 8, 1000, 1), (2000, 16, 150, 2), (77, 255, 9, 3)])
def test_knn_inverse_bit_exact(device, Nq, K, total, seed):
    """CSR transpose: every array identical to the oracle (buckets in (query, k) order).  The third
    case has mean in-degree 213 (> 64: LDS sort path), the fourth 2181 per bucket at K = 255."""
    import pcf_cuda
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, total, (Nq, K)).astype(np.int64)
    idx[rng.random((Nq, K)) < 0.05] = -1
    idx[rng.random((Nq, K)) < 0.02] = total + 5
    inv_n, inv_k, inv_idx = pcf_cuda.compute_knn_inverse(torch.from_numpy(idx)[None].to(device), total)
    wn, wk, wi = O.knn_inverse(idx, total)
    assert inv_n.dtype == torch.int32 and inv_k.dtype == torch.uint8 and inv_idx.dtype == torch.int32
    np.testing.assert_array_equal(inv_idx[0].cpu().numpy(), wi)
    np.testing.assert_array_equal(inv_n[0].cpu().numpy(), wn)
    np.testing.assert_array_equal(inv_k[0].cpu().numpy(), wk)


def test_knn_inverse_one_huge_bucket(device):
    """Every edge points at the same target: 8000 entries in one bucket (rank-by-counting path)."""
    import pcf_cuda
    idx = np.full((500, 16), 3, np.int64)
    inv_n, inv_k, inv_idx = pcf_cuda.compute_knn_inverse(torch.from_numpy(idx)[None].to(device), 10)
    wn, wk, wi = O.knn_inverse(idx, 10)
    np.testing.assert_array_equal(inv_idx[0].cpu().numpy(), wi)
    np.testing.assert_array_equal(inv_n[0].cpu().numpy(), wn)
    np.testing.assert_array_equal(inv_k[0].cpu().numpy(), wk)


def test_knn_inverse_batched_equals_oracle(device):
    """All tables of an iteration in one pass of launches (pcf_hip_knn_inverse_batched): 19 tables (two launch groups of
    16 + 3) of very different sizes -- small K, K = 255, mean in-degree > 64 (LDS sort), one bucket holding every edge
    (rank-by-counting), out-of-range indices, a single edge, and an empty table that takes the single-table path -- each
    bit-identical to the oracle."""
    import pcf_cuda
    rng = np.random.default_rng(11)
    shapes = [(500, 16, 500), (300, 8, 1000), (2000, 16, 150), (77, 255, 9), (1, 1, 1), (4000, 16, 4000), (900, 12, 3000)]
    shapes += [(int(rng.integers(5, 600)), int(rng.integers(1, 20)), int(rng.integers(1, 900))) for _ in range(10)]
    tables, totals = [], []
    for Nq, K, total in shapes:
        idx = rng.integers(0, total, (Nq, K)).astype(np.int64)
        idx[rng.random((Nq, K)) < 0.05] = -1
        idx[rng.random((Nq, K)) < 0.02] = total + 5
        tables.append(idx)
        totals.append(total)
    tables.append(np.full((500, 16), 3, np.int64)); totals.append(10)          # one bucket with 8000 entries
    tables.append(np.zeros((0, 16), np.int64)); totals.append(7)                # empty table
    got = pcf_cuda.compute_knn_inverse_batched([torch.from_numpy(t)[None].to(device) for t in tables], totals)
    assert len(got) == len(tables)
    for (inv_n, inv_k, inv_idx), idx, total in zip(got, tables, totals):
        wn, wk, wi = O.knn_inverse(idx, total)
        assert inv_n.shape == (1, idx.size) and inv_k.shape == (1, idx.size) and inv_idx.shape == (1, total + 1)
        assert inv_n.dtype == torch.int32 and inv_k.dtype == torch.uint8 and inv_idx.dtype == torch.int32
        np.testing.assert_array_equal(inv_idx[0].cpu().numpy(), wi)
        np.testing.assert_array_equal(inv_n[0].cpu().numpy(), wn)
        np.testing.assert_array_equal(inv_k[0].cpu().numpy(), wk)


@pytest.mark.parametrize('K', [1, 5, 16, 24, 40])
def test_knn_bit_exact_small(device, K):
    import pcf_cuda
    rng = np.random.default_rng(K)
    ref = rng.random((900, 3), dtype=np.float32)
    qry = rng.random((700, 3), dtype=np.float32)
    ref[10] = ref[3]                                   # duplicate points: ties resolved by index
    roff, qoff = [0, 300, 900], [0, 500, 700]
    got = pcf_cuda.knn_packed(torch.from_numpy(ref).to(device), torch.from_numpy(qry).to(device),
                              torch.tensor(roff, dtype=torch.int32, device=device),
                              torch.tensor(qoff, dtype=torch.int32, device=device), K)
    np.testing.assert_array_equal(got.cpu().numpy(), O.knn_packed(ref, qry, roff, qoff, K))


def test_knn_golden_and_self_first(device):
    import pcf_cuda
    for tag, K in (('self', 16), ('cross', 16), ('k5', 5)):
        g = load_golden('knn_' + tag)
        off = lambda n: torch.tensor([0, n], dtype=torch.int32, device=device)
        got = pcf_cuda.knn_packed(g['ref'].to(device), g['query'].to(device), off(len(g['ref'])), off(len(g['query'])), K)
        got = got.cpu().numpy()
        dist = g['dist'].numpy()
        untied = np.all(np.diff(dist, axis=1) > 1e-7, axis=1)
        np.testing.assert_array_equal(got[untied], g['idx'].numpy()[untied])
        if tag == 'self':
            np.testing.assert_array_equal(got[:, 0], np.arange(len(got)))   # layers.py:377-378 relies on this


def test_knn_fewer_refs_than_k(device):
    import pcf_cuda
    ref = torch.rand(5, 3)
    got = pcf_cuda.knn_packed(ref.to(device), ref.to(device), torch.tensor([0, 5], dtype=torch.int32, device=device),
                              torch.tensor([0, 5], dtype=torch.int32, device=device), 8).cpu()
    assert (got[:, 5:] == -1).all() and (got[:, :5] >= 0).all()


def test_knn_bit_exact_20k(device):
    from oracle import knn_c
    import pcf_cuda
    rng = np.random.default_rng(42)
    ref = (rng.random((20000, 3), dtype=np.float32) * 4).astype(np.float32)
    roff = [0, 7000, 20000]
    got = pcf_cuda.knn_packed(torch.from_numpy(ref).to(device), torch.from_numpy(ref).to(device),
                              torch.tensor(roff, dtype=torch.int32, device=device),
                              torch.tensor(roff, dtype=torch.int32, device=device), 16)
    np.testing.assert_array_equal(got.cpu().numpy(), knn_c.knn_packed(ref, ref, roff, roff, 16))


def test_gemm_nt(device):
    import pcf_cuda
    g = torch.Generator().manual_seed(0)
    for M, N, Kd in [(1000, 32, 256), (333, 96, 288), (64, 64, 16), (70, 5, 3), (4096, 192, 1536)]:
        a, b, bias = torch.randn(M, Kd, generator=g), torch.randn(N, Kd, generator=g), torch.randn(N, generator=g)
        got = pcf_cuda.gemm_nt(a.to(device), b.to(device), bias.to(device)).cpu()
        want = (a.double() @ b.double().t() + bias.double()).float()
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4 * Kd ** 0.5)


# ---- BASELINE size (N = 80 000, K = 16, Ci = 16, Cm = 16, H = 8): size-independent properties ----
def test_full_size_adjoint_and_linearity(device):
    """<gout, F(x)> == <grad_x, x>, same for w and guidance (F is linear in each), and
    F(a*x1 + b*x2) == a F(x1) + b F(x2); N = 80k is far beyond what the CPU oracle runs in seconds."""
    import pcf_cuda
    N, K, Ci, Cm, H = 80000, 16, 16, 16, 8
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn(1, N, Ci, generator=g).to(device)
    x2 = torch.randn(1, N, Ci, generator=g).to(device)
    idx = torch.randint(0, N, (1, N, K), generator=g).to(device)
    guid = torch.rand(1, N, K, H, generator=g).to(device)
    w = torch.randn(1, N, K, Cm, generator=g).to(device)
    gout = torch.randn(1, N, Ci * Cm, generator=g).to(device)
    out = pcf_cuda.pcf_forward(x, idx, guid, w)
    gx, gg, gw = pcf_cuda.pcf_backward(gout, x, idx, guid, w)
    lhs = (gout.double() * out.double()).sum()
    for grad, arg in ((gx, x), (gg, guid), (gw, w)):
        rhs = (grad.double() * arg.double()).sum()
        assert abs(float(lhs - rhs)) <= 1e-5 * abs(float(lhs)) + 1e-2, (float(lhs), float(rhs))
    mix = pcf_cuda.pcf_forward(0.5 * x - 2.0 * x2, idx, guid, w)
    torch.testing.assert_close(mix, 0.5 * out - 2.0 * pcf_cuda.pcf_forward(x2, idx, guid, w), rtol=1e-4, atol=1e-3)
    # spot rows against the oracle
    rows = torch.tensor([0, 1, 39999, 79999])
    sub = O.pcf_forward(x.cpu(), idx.cpu()[:, rows], guid.cpu()[:, rows], w.cpu()[:, rows])
    torch.testing.assert_close(out.cpu()[:, rows], sub, **TOL)


@pytest.mark.parametrize('shape', PCF_SHAPES[:4])
def test_pcf_backward_csr(device, shape):
    """grad_x by CSR gather-reduce: equal to the oracle, and bitwise reproducible."""
    import pcf_cuda
    B, N, Nout, K, Ci, Cm, H = shape
    x, idx, guid, w, _ = _case(B, N, Nout, K, Ci, 0, Cm, H, seed=sum(shape) + 1)
    d = lambda t: t.to(device)
    gout = torch.randn(B, Nout, Ci * Cm, generator=torch.Generator().manual_seed(7))
    inv = pcf_cuda.compute_knn_inverse(d(idx), N)
    got = pcf_cuda.pcf_backward_csr(d(gout), d(x), *inv, d(idx), d(guid), d(w))
    for a, b in zip(got, O.pcf_backward(gout, x, idx, guid, w)):
        torch.testing.assert_close(a.cpu(), b, **TOL)
    again = pcf_cuda.pcf_backward_csr(d(gout), d(x), *inv, d(idx), d(guid), d(w))
    assert torch.equal(again[0], got[0])


def _surface(n, rng, side=20.0):
    xy = rng.random((n, 2), dtype=np.float32) * side
    z = (0.35 * np.sin(1.1 * xy[:, 0]) + 0.25 * np.cos(0.7 * xy[:, 1])).astype(np.float32)
    return np.concatenate([xy, z[:, None]], 1).astype(np.float32)


@pytest.mark.parametrize('case', ['volume', 'surface', 'cross', 'plane', 'line', 'duplicates', 'tiny_segments', 'k40'])
def test_knn_grid_equals_bruteforce(device, case):
    """The grid and the wave-per-query engines return exactly what the brute-force engine returns (which the oracle pins):
    volumes, folded sheets, exactly planar / collinear clouds (degenerate grid axes), heavy
    duplicates (ties by index), queries outside the reference box, tiny samples inside a packed batch."""
    import pcf_cuda
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    K = 16
    if case == 'volume':
        ref = rng.random((30000, 3), dtype=np.float32); qry = ref; roff = qoff = [0, 12000, 30000]
    elif case == 'surface':
        ref = _surface(40000, rng); qry = ref; roff = qoff = [0, 40000]
    elif case == 'cross':
        ref = _surface(9000, rng); qry = _surface(30000, rng, side=24.0) - 2.0; roff = [0, 4000, 9000]; qoff = [0, 10000, 30000]
    elif case == 'plane':
        ref = rng.random((20000, 3), dtype=np.float32); ref[:, 2] = 0.5; qry = ref; roff = qoff = [0, 20000]
    elif case == 'line':
        ref = np.zeros((6000, 3), np.float32); ref[:, 0] = rng.random(6000, dtype=np.float32) * 100; qry = ref; roff = qoff = [0, 6000]
    elif case == 'duplicates':
        base = rng.random((500, 3), dtype=np.float32); ref = np.repeat(base, 20, 0); rng.shuffle(ref); qry = ref[:3000]; roff = [0, 10000]; qoff = [0, 3000]
    elif case == 'tiny_segments':
        ref = rng.random((5000, 3), dtype=np.float32); qry = ref; roff = qoff = [0, 3, 20, 4000, 4000, 5000]
    else:
        ref = rng.random((8000, 3), dtype=np.float32); qry = rng.random((5000, 3), dtype=np.float32) * 1.5 - 0.25; roff = [0, 8000]; qoff = [0, 5000]; K = 40
    t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a), dtype=dt).to(device)
    args = (t(ref), t(qry), t(roff, torch.int32), t(qoff, torch.int32), K)
    brute = pcf_cuda.knn_packed(*args, method='brute').cpu().numpy()
    grid = pcf_cuda.knn_packed(*args, method='grid').cpu().numpy()
    np.testing.assert_array_equal(grid, brute)
    wave = pcf_cuda.knn_packed(*args, method='wave').cpu().numpy()
    np.testing.assert_array_equal(wave, brute)


def test_knn_grid_bit_exact_vs_c_oracle_80k(device):
    """BASELINE size: 80 000 points, K = 16, against the C oracle (oracle/knn_ref.c), ~15 s of CPU."""
    from oracle import knn_c
    import pcf_cuda
    rng = np.random.default_rng(80)
    ref = rng.random((80000, 3), dtype=np.float32)
    off = torch.tensor([0, 80000], dtype=torch.int32, device=device)
    got = pcf_cuda.knn_packed(torch.from_numpy(ref).to(device), torch.from_numpy(ref).to(device), off, off, 16)
    sub = np.arange(0, 80000, 16)                       # every 16th query keeps the oracle at seconds
    want = knn_c.knn_packed(ref, ref[sub], [0, 80000], [0, len(sub)], 16)
    np.testing.assert_array_equal(got.cpu().numpy()[sub], want)


# ---- attention arithmetic of the ablation layers (csrc/attention_ops.hip) against plain torch ------------------------
@pytest.mark.parametrize('B,M,K,C,J', [(1, 300, 16, 64, 8), (2, 77, 5, 48, 48), (1, 50, 64, 32, 4)])
def test_softmax_aggregate_against_torch(device, B, M, K, C, J):
    import pcf_fused
    g = torch.Generator().manual_seed(B * 1000 + M)
    v = torch.randn(B, M, K, C, generator=g).to(device).requires_grad_(True)
    lg = (torch.randn(B, M, K, J, generator=g) * 2).to(device).requires_grad_(True)
    out = pcf_fused.softmax_aggregate(v, lg)
    up = torch.randn(B, M, C, generator=g).to(device)
    out.backward(up)
    vr, lr = v.detach().clone().requires_grad_(True), lg.detach().clone().requires_grad_(True)
    w = torch.softmax(lr, dim=2)
    want = (vr.view(B, M, K, C // J, J) * w.unsqueeze(3)).sum(2).view(B, M, C)          # layers.py:523-527
    want.backward(up)
    torch.testing.assert_close(out, want, **TOL)
    torch.testing.assert_close(v.grad, vr.grad, **TOL)
    torch.testing.assert_close(lg.grad, lr.grad, **TOL)


@pytest.mark.parametrize('B,N,K,H,D', [(1, 200, 16, 8, 16), (2, 50, 7, 4, 5)])
def test_qk_score_against_torch(device, B, N, K, H, D):
    import pcf_fused
    g = torch.Generator().manual_seed(N)
    q = torch.randn(B, N, K, H, D, generator=g).to(device).requires_grad_(True)
    key = torch.randn(B, N, H, D, generator=g).to(device).requires_grad_(True)
    s = pcf_fused.qk_score(q, key, D ** -0.5)
    up = torch.randn(B, N, K, H, generator=g).to(device)
    s.backward(up)
    qr, kr = q.detach().clone().requires_grad_(True), key.detach().clone().requires_grad_(True)
    want = torch.sigmoid((qr * kr[:, :, None]).sum(-1) * D ** -0.5)                     # layers.py:108-113
    want.backward(up)
    torch.testing.assert_close(s, want, **TOL)
    torch.testing.assert_close(q.grad, qr.grad, **TOL)
    torch.testing.assert_close(key.grad, kr.grad, **TOL)


@pytest.mark.parametrize('shape,C', [((1, 500, 16), 64), ((3, 41), 20), ((1, 9000), 130)])
def test_layer_norm_against_torch(device, shape, C):
    import pcf_fused
    g = torch.Generator().manual_seed(C)
    ln = torch.nn.LayerNorm(C).to(device)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(C, generator=g) + 0.5)
        ln.bias.copy_(torch.randn(C, generator=g))
    x = (torch.randn(*shape, C, generator=g) * 3 + 1).to(device).requires_grad_(True)
    y = pcf_fused.layer_norm(x, ln)
    up = torch.randn(*shape, C, generator=g).to(device)
    y.backward(up)
    got = (x.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone())
    ln.zero_grad()
    xr = x.detach().clone().requires_grad_(True)
    want = ln(xr)
    want.backward(up)
    torch.testing.assert_close(y, want, **TOL)
    torch.testing.assert_close(got[0], xr.grad, **TOL)
    torch.testing.assert_close(got[1], ln.weight.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(got[2], ln.bias.grad, rtol=1e-3, atol=1e-3)


# ---- point-level Linear + BatchNorm chain kernels (csrc/fused_linear.hip) against torch autograd ---------------------
def _torch_chain(x, W1, b1, g1, be1, W2, b2, g2, be2, res, act1, act2, eps=1e-5):
    """y = act2(BN2(act1(BN1(x W1^T + b1)) W2^T + b2) + res) with batch statistics, in float64."""
    import torch.nn.functional as F
    acts = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.1), 3: torch.sigmoid}

    def bn(z, g, b):
        return (z - z.mean(0)) * torch.rsqrt(z.var(0, unbiased=False) + eps) * g + b

    z1 = x @ W1.t() + b1
    y1 = acts[act1](bn(z1, g1, be1))
    z2 = y1 @ W2.t() + b2
    return acts[act2](bn(z2, g2, be2) + res), z1, z2


@pytest.mark.parametrize('R,C0,C1,C2,act1,act2', [(5000, 64, 16, 32, 2, 0), (777, 256, 32, 64, 1, 2), (1300, 6, 20, 70, 2, 2),
                                                  (33, 16, 8, 8, 1, 1), (3000, 96, 48, 192, 1, 2)])
@pytest.mark.parametrize('split_k,finish', [(-1, True), (1, True), (-1, False), (1, False)])
def test_fused_linear_chain_kernels_against_torch(device, R, C0, C1, C2, act1, act2, split_k, finish):
    """Two chained Linear+BatchNorm layers through the four C-ABI entry points (forward with the producer's BatchNorm in the
    A-tile loader and a side output, top-of-chain statistics, dz-prologue input / weight gradients with the producer's
    statistics in the epilogue): outputs, running statistics, input gradient and all parameter gradients against float64
    torch autograd, 1e-3 of the scale."""
    import pcf_fused as PF
    import pcf_cuda
    pcf_cuda.set_flin_split_k(split_k)         # 1: the 32 x 32 split-K kernels wherever K >= 64
    pcf_cuda.set_flin_finish(finish)           # statistics combined by a launch of their own / by the last workgroup (tickets)
    g = torch.Generator().manual_seed(R + C2)
    r = lambda *s: torch.randn(*s, generator=g)
    x, W1, b1, W2, b2 = r(R, C0), r(C1, C0) / C0 ** 0.5, r(C1) * 0.1, r(C2, C1) / C1 ** 0.5, r(C2) * 0.1
    g1, be1, g2, be2, res, up = torch.rand(C1, generator=g) + 0.5, r(C1) * 0.2, torch.rand(C2, generator=g) + 0.5, r(C2) * 0.2, r(R, C2), r(R, C2)
    d = lambda t: t.to(device).contiguous()
    bn1, bn2 = torch.nn.BatchNorm1d(C1).to(device), torch.nn.BatchNorm1d(C2).to(device)
    with torch.no_grad():
        bn1.weight.copy_(g1); bn1.bias.copy_(be1); bn2.weight.copy_(g2); bn2.bias.copy_(be2)
    dev = device
    s = PF._stream(dev)
    xd, W1d, b1d, W2d, b2d, resd, upd = d(x), d(W1), d(b1), d(W2), d(b2), d(res), d(up)
    with PF._guard(dev):
        z1, cst1 = PF._flin_forward(xd, None, 0, None, W1d, b1d, bn1, 0.1, s, dev)
        y1 = torch.empty(R, C1, device=dev)
        z2, cst2 = PF._flin_forward(z1, cst1, act1, y1, W2d, b2d, bn2, 0.1, s, dev)
        out = torch.empty_like(z2)
        PF._call(PF._bnact_fwd, PF._ptr(z2), PF._ptr(resd), R, C2, cst2[2].data_ptr(), cst2[3].data_ptr(), PF._ptr(bn2.weight),
                 PF._ptr(bn2.bias), act2, PF._ptr(out), s)
        # backward
        f32 = dict(dtype=torch.float32, device=dev)
        dg2, dbe2, db2 = torch.empty(C2, **f32), torch.empty(C2, **f32), torch.empty(C2, **f32)
        dg1, dbe1, db1 = torch.empty(C1, **f32), torch.empty(C1, **f32), torch.empty(C1, **f32)
        gg = torch.empty_like(z2)
        ws, nbytes = PF._ws(dev, R, C2, C2)
        PF._call(PF._bn_bwd_stats, PF._ptr(upd), PF._ptr(z2), PF._ptr(resd), PF._ptr(cst2), act2, R, C2, PF._ptr(gg), PF._ptr(dg2),
                 PF._ptr(dbe2), PF._ptr(db2), ws.data_ptr(), nbytes, PF._tickets(dev).data_ptr(), s)
        wg = PF._WeightGrads(dev, s)
        dW2 = wg.add(gg, z2, cst2, 0, z1, cst1, act1)
        dy1 = PF._flin_bwd_input(gg, z2, cst2, 0, W2d, None, z1, cst1, act1, (dg1, dbe1, db1), s, dev)
        dW1 = wg.add(dy1, z1, cst1, act1, xd, None, 0)
        dx = PF._flin_bwd_input(dy1, z1, cst1, act1, W1d, None, None, None, 0, None, s, dev)
        wg.finish()
    t = lambda v: v.double().requires_grad_(True)
    X, A1, B1, G1, E1, A2, B2, G2, E2, RS = t(x), t(W1), t(b1), t(g1), t(be1), t(W2), t(b2), t(g2), t(be2), t(res)
    want, rz1, rz2 = _torch_chain(X, A1, B1, G1, E1, A2, B2, G2, E2, RS, act1, act2)
    want.backward(up.double())

    def close(got, ref, what, tol=1e-3):
        ref = ref.float()
        torch.testing.assert_close(got.cpu(), ref, rtol=tol, atol=tol * max(1.0, float(ref.abs().max())), msg=lambda m: f'{what}: {m}')

    close(z1, rz1.detach(), 'z1'); close(z2, rz2.detach(), 'z2'); close(out, want.detach(), 'out')
    acts = {0: lambda v: v, 1: torch.relu, 2: lambda v: torch.nn.functional.leaky_relu(v, 0.1), 3: torch.sigmoid}
    z1d = rz1.detach()
    close(y1, acts[act1]((z1d - z1d.mean(0)) * torch.rsqrt(z1d.var(0, unbiased=False) + 1e-5) * g1.double() + be1.double()), 'side output')
    close(bn1.running_mean, 0.1 * z1d.mean(0), 'running_mean 1')
    close(bn2.running_var, 0.9 + 0.1 * rz2.detach().var(0, unbiased=True), 'running_var 2')
    close(gg, RS.grad, 'g (= residual gradient)')
    close(dx, X.grad, 'dx'); close(dW1, A1.grad, 'dW1'); close(dW2, A2.grad, 'dW2')
    close(dg1, G1.grad, 'dgamma1'); close(dbe1, E1.grad, 'dbeta1'); close(dg2, G2.grad, 'dgamma2'); close(dbe2, E2.grad, 'dbeta2')
    assert float(db1.abs().max()) == 0.0 and float(db2.abs().max()) == 0.0          # bias in front of a batch-statistics BatchNorm
    assert float(PF._tickets(dev).abs().sum()) == 0.0                                 # the kernels leave their tickets zeroed
    pcf_cuda.set_flin_split_k(-1)
    pcf_cuda.set_flin_finish(True)


def test_fused_linear_rejects_bad_arguments(device):
    import pcf_fused as PF
    dev = device
    x = torch.randn(10, 8, device=dev)
    W = torch.randn(4, 8, device=dev)
    bn = torch.nn.BatchNorm1d(4).to(dev)
    with pytest.raises(RuntimeError, match='workspace|statistics'):
        PF._call(PF._flin_fwd, PF._ptr(x), 10, 8, None, 0, None, PF._ptr(W), None, 4, PF._ptr(torch.empty(10, 4, device=dev)),
                 PF._ptr(torch.empty(6, 4, device=dev)), PF._ptr(bn.weight), PF._ptr(bn.bias), None, None, 1e-5, 0.1, None, 0, None,
                 PF._stream(dev))
    # zero rows: nothing to do, no launch
    z, cst = PF._flin_forward(torch.empty(0, 8, device=dev), None, 0, None, W, None, None, 0.0, PF._stream(dev), dev)
    assert z.shape == (0, 4)


# ---- PCFLayer head / tail as row chains (csrc/point_chain.hip) against float64 torch autograd --------------------------
def _bn64(z, g, b, eps=1e-5):
    return (z - z.mean(0)) * torch.rsqrt(z.var(0, unbiased=False) + eps) * g + b


def _close(got, ref, what, tol=1e-3):
    ref = ref.float()
    torch.testing.assert_close(got.cpu(), ref, rtol=tol, atol=tol * max(1.0, float(ref.abs().max())), msg=lambda m: f'{what}: {m}')


@pytest.mark.parametrize('R', [17, 4099, 80000])
def test_point_head_row_chain_against_torch(device, R):
    """unary1 -> guidance_unary -> gathered half of the first guidance layer (layers.py:335, 369, 372) in the three-pass row chain:
    fx, u, running statistics, dx and all ten parameter gradients against float64 autograd, 1e-3 of the scale."""
    import pcf_fused as PF
    import torch.nn.functional as F
    cin, mid, G = 64, 16, 32
    assert PF.point_head_chain_supported(cin, mid, G)
    g = torch.Generator().manual_seed(R)
    r = lambda *s: torch.randn(*s, generator=g)
    x, W1, b1, W2, b2, Wa = r(R, cin), r(mid, cin) / cin ** 0.5, r(mid) * 0.1, r(G, mid) / mid ** 0.5, r(G) * 0.1, r(8, G) / G ** 0.5
    g1, be1, g2, be2 = torch.rand(mid, generator=g) + 0.5, r(mid) * 0.2, torch.rand(G, generator=g) + 0.5, r(G) * 0.2
    upf, upu = r(R, mid), r(R, 8)
    bn1, bn2 = torch.nn.BatchNorm1d(mid).to(device), torch.nn.BatchNorm1d(G).to(device)
    P = [t.to(device).requires_grad_(True) for t in (x, Wa, W2, b2, g2, be2, W1, b1, g1, be1)]
    fx, u = PF._PointHeadChain.apply((bn1, bn2), *P)
    torch.autograd.backward([fx, u], [upf.to(device), upu.to(device)])
    t = lambda v: v.double().requires_grad_(True)
    Q = [t(v) for v in (x, Wa, W2, b2, g2, be2, W1, b1, g1, be1)]
    X, A, A2, B2, G2, E2, A1, B1, G1, E1 = Q
    z1 = X @ A1.t() + B1
    rfx = F.leaky_relu(_bn64(z1, G1, E1), 0.1)
    z2 = rfx @ A2.t() + B2
    ru = _bn64(z2, G2, E2) @ A.t()
    torch.autograd.backward([rfx, ru], [upf.double(), upu.double()])
    _close(fx, rfx.detach(), 'fx'); _close(u, ru.detach(), 'u')
    _close(bn1.running_mean, 0.1 * z1.detach().mean(0), 'running_mean 1')
    _close(bn2.running_var, 0.9 + 0.1 * z2.detach().var(0, unbiased=True), 'running_var 2')
    names = ['dx', 'dWa', 'dW2', 'db2', 'dgamma2', 'dbeta2', 'dW1', 'db1', 'dgamma1', 'dbeta1']
    for n, a, b in zip(names, P, Q):
        _close(a.grad, b.grad, n)
    assert float(PF._tickets(device).abs().sum()) == 0.0


@pytest.mark.parametrize('R,finish_launch', [(17, False), (4099, False), (80000, False), (4099, True)])
def test_point_tail_row_chain_against_torch(device, R, finish_launch):
    """linear -> unary2 -> + shortcut -> LeakyReLU (layers.py:393-414) in the row chain: output, running statistics, the gradients
    of the aggregate, of the shortcut and of all eight parameters against float64 autograd, 1e-3 of the scale; and the chain
    agrees with the layer-by-layer contraction kernels it replaces."""
    import pcf_fused as PF
    import pcf_cuda
    import torch.nn.functional as F
    pcf_cuda.set_row_chain_finish(finish_launch)      # statistics by a launch of their own instead of the last workgroup
    ca, ch, co = 256, 32, 64
    assert PF.point_tail_chain_supported(ca, ch, co)
    g = torch.Generator().manual_seed(R + 1)
    r = lambda *s: torch.randn(*s, generator=g)
    agg, sc, W3, b3, W4, b4 = r(R, ca), r(R, co), r(ch, ca) / ca ** 0.5, r(ch) * 0.1, r(co, ch) / ch ** 0.5, r(co) * 0.1
    g3, be3, g4, be4 = torch.rand(ch, generator=g) + 0.5, r(ch) * 0.2, torch.rand(co, generator=g) + 0.5, r(co) * 0.2
    up = r(R, co)
    vals = (agg, sc, W3, b3, g3, be3, W4, b4, g4, be4)
    results = {}
    for fn in (PF._PointTailChain, PF._PointTail):
        bn3, bn4 = torch.nn.BatchNorm1d(ch).to(device), torch.nn.BatchNorm1d(co).to(device)
        with torch.no_grad():
            bn3.weight.copy_(g3); bn3.bias.copy_(be3); bn4.weight.copy_(g4); bn4.bias.copy_(be4)
        P = [t.to(device).requires_grad_(True) for t in vals]
        P[4], P[5], P[8], P[9] = bn3.weight, bn3.bias, bn4.weight, bn4.bias      # the layer-by-layer form reads the modules
        out = fn.apply((bn3, bn4), *P)
        out.backward(up.to(device))
        results[fn] = (out, P, bn3, bn4)
    out, P, bn3, bn4 = results[PF._PointTailChain]
    t = lambda v: v.double().requires_grad_(True)
    Q = [t(v) for v in vals]
    X, S, A3, B3, G3, E3, A4, B4, G4, E4 = Q
    z3 = X @ A3.t() + B3
    z4 = F.relu(_bn64(z3, G3, E3)) @ A4.t() + B4
    want = F.leaky_relu(_bn64(z4, G4, E4) + S, 0.1)
    want.backward(up.double())
    _close(out.detach(), want.detach(), 'out')
    _close(bn3.running_mean, 0.1 * z3.detach().mean(0), 'running_mean 3')
    _close(bn4.running_var, 0.9 + 0.1 * z4.detach().var(0, unbiased=True), 'running_var 4')
    names = ['dagg', 'dshortcut', 'dW3', 'db3', 'dgamma3', 'dbeta3', 'dW4', 'db4', 'dgamma4', 'dbeta4']
    for n, a, b in zip(names, P, Q):
        _close(a.grad, b.grad, n)
    out2, P2, _, _ = results[PF._PointTail]
    _close(out, out2.detach().cpu(), 'chain vs layer-by-layer', tol=1e-4)
    for n, a, b in zip(names, P, P2):
        if n == 'dagg':          # a ReLU whose argument rounds to opposite sides of zero in the two forms changes a whole row
            bad = ((a.grad - b.grad).abs() > 2e-4 * max(1.0, float(b.grad.abs().max()))).any(dim=1)
            assert int(bad.sum()) <= max(2, R // 20000), f'{int(bad.sum())} rows of dagg differ between the two forms'
            continue
        _close(a.grad, b.grad.cpu(), n + ' (chain vs layer-by-layer)', tol=2e-4 if R < 80000 else 5e-3)   # the same flipped rows
    pcf_cuda.set_row_chain_finish(False)
    assert float(PF._tickets(device).abs().sum()) == 0.0
