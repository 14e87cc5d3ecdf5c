"""CPU: the oracle (oracle/pcf_oracle.py) against the golden vectors the reference produced.

This is what pins the oracle: outputs and autograd gradients of the reference's
pure-PyTorch layers, and the tensors at the pcf_cuda operator boundary captured inside them.
Tolerance: 1e-4 absolute/relative in fp32 (same maths, different op order); the HIP tests
then hold the kernels to BASELINE's 1e-3 against this oracle.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, split
from oracle import pcf_oracle as O

TOL = dict(rtol=1e-4, atol=1e-4)


def _params(g, training=True):
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k)
          for k, v in split(g, 'sd.').items()}
    return sd, O.Params(sd, '', training)


def _check_param_grads(g, sd):
    want = split(g, 'gsd.')
    assert want
    for k, v in want.items():
        got = sd[k].grad
        assert got is not None, k
        tol = TOL
        if k.endswith('.c.bias') and k[:-len('c.bias')] + 'bn.weight' in sd:
            # a bias in front of a batch-stat BatchNorm has an analytically ZERO gradient; both
            # sides hold fp32 cancellation noise (~1e-4 * sum|g|), so only bound its size.
            tol = dict(rtol=0, atol=2e-3)
        torch.testing.assert_close(got, v, **tol, msg=lambda m, k=k: f'{k}: {m}')


@pytest.mark.parametrize('name', ['pcf_self_64', 'pcf_self_32_64', 'pcf_strided', 'pcf_qk_self', 'pcf_qk_strided',
                                  'pcf_ln_self', 'pcf_ln_strided'])
def test_pcf_layer(name):
    g = load_golden(name)
    a = split(g, 'in.')
    sd, P = _params(g)
    feats = a['dense_feats'].clone().requires_grad_(True)
    out, wn = O.pcf_layer(P, a['dense_xyz'], feats, a['nei_inds'], a['dense_xyz_norm'],
                          a.get('sparse_xyz'), a.get('sparse_xyz_norm'),
                          num_heads=int(g['meta.num_heads']))
    torch.testing.assert_close(wn, g['out.wn_in'], **TOL)
    torch.testing.assert_close(out, g['out.new_feat'], **TOL)
    out.backward(g['gup'])
    torch.testing.assert_close(feats.grad, g['gin.dense_feats'], **TOL)
    _check_param_grads(g, sd)


@pytest.mark.parametrize('name', ['pcf_self_64', 'pcf_self_32_64', 'pcf_strided'])
def test_pcf_operator_boundary(name):
    """pcf_forward/pcf_backward against tensors captured around the aggregate inside PCFLayer."""
    g = load_golden(name)
    x, idx = g['cap.fx'], g['in.nei_inds']
    out = O.pcf_forward(x, idx, g['cap.score'], g['cap.w'])
    torch.testing.assert_close(out, g['cap.agg'], **TOL)
    gx, gg, gw = O.pcf_backward(g['gcap.agg'], x, idx, g['cap.score'], g['cap.w'])
    torch.testing.assert_close(gg, g['gcap.score'], **TOL)
    torch.testing.assert_close(gw, g['gcap.w'], **TOL)
    # grad_x has no isolated golden (fx also feeds the guidance branch): check the closed form
    # against autograd of the oracle forward instead.
    xr = x.clone().requires_grad_(True)
    O.pcf_forward(xr, idx, g['cap.score'], g['cap.w']).backward(g['gcap.agg'])
    torch.testing.assert_close(gx, xr.grad, **TOL)


def test_pointconv_single():
    g = load_golden('pointconv_single')
    a = split(g, 'in.')
    sd, P = _params(g)
    feats = a['dense_feats'].clone().requires_grad_(True)
    out, wn = O.pointconv_layer(P, a['dense_xyz'], feats, a['nei_inds'], use_vi=False, use_pe=False)
    torch.testing.assert_close(out, g['out.new_feat'], **TOL)
    out.backward(g['gup'])
    torch.testing.assert_close(feats.grad, g['gin.dense_feats'], **TOL)
    _check_param_grads(g, sd)
    # operator boundary: pconv_linear_forward/backward, Ca = 0
    idx, w = a['nei_inds'], g['cap.w']
    add = torch.zeros(1, idx.shape[1], idx.shape[2], 0)
    lw, lb = g['sd.linear.weight'], g['sd.linear.bias']
    lin, p = O.pconv_linear_forward(a['dense_feats'], idx, w, add, lw, lb)
    torch.testing.assert_close(p, g['cap.agg'], **TOL)
    torch.testing.assert_close(lin, g['cap.lin'], **TOL)
    gx, gw, ga, glw, glb = O.pconv_linear_backward(g['gcap.lin'], a['dense_feats'], idx, w, add, lw, p)
    torch.testing.assert_close(gx, g['gin.dense_feats'], **TOL)      # feats feed only the aggregate here
    torch.testing.assert_close(gw, g['gcap.w'], **TOL)
    torch.testing.assert_close(glw, g['gsd.linear.weight'], **TOL)
    torch.testing.assert_close(glb, g['gsd.linear.bias'], **TOL)
    assert ga.shape[-1] == 0


def test_pointconv_vi_pe():
    g = load_golden('pointconv_vi_pe')
    a = split(g, 'in.')
    sd, P = _params(g)
    feats = a['dense_feats'].clone().requires_grad_(True)
    out, wn = O.pointconv_layer(P, a['dense_xyz'], feats, a['nei_inds'], a['dense_xyz_norm'],
                                use_vi=True, use_pe=True)
    torch.testing.assert_close(wn, g['out.wn_in'], **TOL)
    torch.testing.assert_close(out, g['out.new_feat'], **TOL)
    out.backward(g['gup'])
    torch.testing.assert_close(feats.grad, g['gin.dense_feats'], **TOL)
    _check_param_grads(g, sd)
    # operator boundary with Ca = 12 (additional = VI features)
    idx = a['nei_inds']
    p = O.pconv_forward(a['dense_feats'], idx, g['cap.w'], g['out.wn_in'])
    torch.testing.assert_close(p, g['cap.agg'], **TOL)
    gx, gw, ga = O.pconv_backward(g['gcap.agg'], a['dense_feats'], idx, g['cap.w'], g['out.wn_in'])
    torch.testing.assert_close(gx, g['gin.dense_feats'], **TOL)
    torch.testing.assert_close(gw, g['gcap.w'], **TOL)


def test_stride_pe():
    g = load_golden('stride_pe')
    a = split(g, 'in.')
    sd, P = _params(g)
    feats = a['dense_feats'].clone().requires_grad_(True)
    out, wn = O.pointconv_stride_pe_layer(P, a['dense_xyz'], feats, a['nei_inds'], a['dense_xyz_norm'],
                                          a['sparse_xyz'], a['sparse_xyz_norm'])
    torch.testing.assert_close(out, g['out.new_feat'], **TOL)
    out.backward(g['gup'])
    torch.testing.assert_close(feats.grad, g['gin.dense_feats'], **TOL)
    _check_param_grads(g, sd)
    p = O.pconv_forward(g['cap.fx'], a['nei_inds'], g['cap.w'], g['cap.pe'])
    torch.testing.assert_close(p, g['cap.agg'], **TOL)
    gx, gw, ga = O.pconv_backward(g['gcap.agg'], g['cap.fx'], a['nei_inds'], g['cap.w'], g['cap.pe'])
    torch.testing.assert_close(gw, g['gcap.w'], **TOL)
    torch.testing.assert_close(ga, g['gcap.pe'], **TOL)
    torch.testing.assert_close(gx, g['gcap.fx'], **TOL)


def test_transpose_pe():
    g = load_golden('transpose_pe')
    a = split(g, 'in.')
    sd, P = _params(g)
    feats = a['sparse_feats'].clone().requires_grad_(True)
    skip = a['dense_feats'].clone().requires_grad_(True)
    out, wn = O.pointconv_transpose_pe_layer(P, a['sparse_xyz'], feats, a['nei_inds'], a['sparse_xyz_norm'],
                                             a['dense_xyz'], a['dense_xyz_norm'], skip)
    torch.testing.assert_close(out, g['out.new_feat'], **TOL)
    out.backward(g['gup'])
    torch.testing.assert_close(feats.grad, g['gin.sparse_feats'], **TOL)
    torch.testing.assert_close(skip.grad, g['gin.dense_feats'], **TOL)
    _check_param_grads(g, sd)
    # Cm = 1, Ci = 128, Ca = 16, N < Nout
    p = O.pconv_forward(a['sparse_feats'], a['nei_inds'], g['cap.w'], g['cap.pe'])
    torch.testing.assert_close(p, g['cap.agg'], **TOL)
    gx, gw, ga = O.pconv_backward(g['gcap.agg'], a['sparse_feats'], a['nei_inds'], g['cap.w'], g['cap.pe'])
    torch.testing.assert_close(gx, g['gin.sparse_feats'], **TOL)
    torch.testing.assert_close(gw, g['gcap.w'], **TOL)
    torch.testing.assert_close(ga, g['gcap.pe'], **TOL)


def test_vi_transform():
    g = load_golden('vi_transform')
    xyz, nrm, idx = g['xyz'], g['nrm'], g['idx']
    rel = xyz[idx] - xyz[:, None]
    vi = O.vi_features(rel[None], nrm[idx][None], nrm[None])[0]
    assert torch.isfinite(vi).all()
    torch.testing.assert_close(vi, g['vi'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('tag,K', [('self', 16), ('cross', 16), ('k5', 5)])
def test_knn_matches_kdtree_where_untied(tag, K):
    """The reference holds no kNN fixture (parity unpinned upstream).  Definition pinned here:
    brute force agrees with sklearn KDTree (the reference's CPU default) on every query whose
    K-th and (K+1)-th distances differ."""
    g = load_golden('knn_' + tag)
    got = O.knn_bruteforce(g['ref'].numpy(), g['query'].numpy(), K)
    d = g['dist'].numpy()
    untied = np.all(np.diff(d, axis=1) > 1e-7, axis=1)
    assert untied.mean() > 0.9
    np.testing.assert_array_equal(got[untied], g['idx'].numpy()[untied])


def test_knn_inverse_small_known_answer():
    idx = np.array([[0, 1], [0, 2], [2, -1], [1, 7]], np.int64)     # -1 and 7 are out of range
    inv_n, inv_k, inv_idx = O.knn_inverse(idx, 3)
    assert inv_idx.tolist() == [0, 2, 4, 6]
    assert inv_n.tolist() == [0, 1, 0, 3, 1, 2, 0, 0]
    assert inv_k.tolist() == [0, 0, 1, 0, 1, 0, 0, 0]


@pytest.mark.parametrize('name', ['ptl_self', 'ptl_strided'])
def test_point_transformer_layer(name):
    """The ablation block (layers.py:419-539) as the reference's PyTorch code computed it: output, feature gradient and
    every parameter gradient."""
    g = load_golden(name)
    a = split(g, 'in.')
    sd, P = _params(g)
    feats = a['feats'].clone().requires_grad_(True)
    out = O.point_transformer_layer(P, a['xyz'], feats, a['nei_ind'], a.get('sparse_xyz'), share_planes=8)
    torch.testing.assert_close(out, g['out.new_feat'], **TOL)
    out.backward(g['gup'])
    torch.testing.assert_close(feats.grad, g['gin.feats'], **TOL)
    _check_param_grads(g, sd)


def _model_table(g, dtype=torch.float32):
    """cfg, {name: tensor} of synthetic parameters + default buffers for the model of fixture g (shapes from the build's
    own model definition, constructed on the CPU: only its forward needs the GPU)."""
    import model_fixture as MF
    import pcf_model
    cfg = MF.model_cfg(g)
    c2 = pcf_model.Config(cfg)
    c2.PCONV_OPT, c2.USE_CUDA_KERNEL = False, True
    net = pcf_model.PointConvFormer_Segmentation(c2)
    shapes = {k: tuple(v.shape) for k, v in net.named_parameters()}
    assert sorted(shapes) == sorted(MF.reference_parameter_names(g))
    assert sum(p.numel() for p in net.parameters()) == int(g['meta.n_params'])
    table = {k: v.to(dtype).requires_grad_(True) for k, v in MF.synthetic_state(shapes).items()}
    for k, b in net.named_buffers():
        table[k] = b.clone().to(dtype) if b.is_floating_point() else b.clone()
    return cfg, shapes, table


@pytest.mark.parametrize('tag', ['lite', '10cm', '2cm'])
def test_whole_model_at_real_widths(tag):
    """oracle.segmentation_model against the reference model at the real widths of the BASELINE YAMLs
    (tests/golden/make_golden_models.py), both in float64 so that the comparison is not drowned in the rounding
    amplification of the 29-layer backward: logits, feature gradient, every parameter gradient (whole or sampled) to
    1e-6 of the scale; then in float32 the logits to 5e-4.  The 2cm fixture runs with drop_path_rate 0.2 under the
    recorded keep mask."""
    import model_fixture as MF
    g = MF.load(tag)
    cfg, shapes, table = _model_table(g, torch.float64)
    feats, pcs, es, ef, ep, nrms = MF.inputs(g)
    feats = feats.detach().double().requires_grad_(True)
    out = O.segmentation_model(O.Params(table, '', True), cfg, feats, [p.double() for p in pcs], es, ef, ep,
                               [n.double() for n in nrms], drop_scales=MF.drop_factors(g))
    torch.testing.assert_close(out.float(), g['out'], rtol=1e-5, atol=1e-5)
    out.backward(g['gup'].double())
    torch.testing.assert_close(feats.grad.float(), g['gin.features'], rtol=1e-5, atol=1e-5 * float(g['gin.features'].abs().max()))
    bad = MF.bad_parameter_grads(((k, table[k].grad) for k in shapes), g, rtol=1e-5, atol=1e-5, noise_mult=0.0)
    assert not bad, bad[:8]
    cfg, shapes, table = _model_table(g, torch.float32)
    feats, pcs, es, ef, ep, nrms = MF.inputs(g)
    out = O.segmentation_model(O.Params(table, '', True), cfg, feats, pcs, es, ef, ep, nrms, drop_scales=MF.drop_factors(g))
    torch.testing.assert_close(out, g['out'], rtol=5e-4, atol=5e-4)


@pytest.mark.parametrize('tag', ['lite', '10cm', '2cm'])
def test_model_blocks_at_real_widths(tag):
    """Every block of the model on its own (parameters and neighbourhoods of the model, synthetic features and upstream
    gradient): the oracle's layer functions in float32 against the reference block's float64 output, input gradients
    and parameter gradients, 1e-3 (BASELINE's bar; in practice ~1e-5)."""
    import model_fixture as MF
    g = MF.load(tag)
    cfg, shapes, table = _model_table(g)
    _, pcs, es, ef, ep, nrms = MF.inputs(g)
    drops = MF.drop_factors(g)
    H, pe = cfg.num_heads, cfg.USE_PE
    for name, kind, lin, lout, cin, cout in MF.block_plan(cfg):
        feats, skip, up, edges = MF.block_case(g, name, kind, lin, lout, cin, cout)
        P = O.Params(table, name + '.', True)
        guided = P.has('guidance_unary.mlp.c.weight')
        if kind == 'pointconv':
            out, _ = O.pointconv_layer(P, pcs[lin], feats, edges, nrms[lin], use_vi=True, use_pe=pe)
        elif kind == 'up':
            out, _ = O.pointconv_transpose_pe_layer(P, pcs[lin], feats, edges, nrms[lin], pcs[lout], nrms[lout], skip, use_pe=pe)
        else:
            sp = (pcs[lout], nrms[lout]) if kind == 'down' else (None, None)
            fn = O.pcf_layer if guided else O.pointconv_stride_pe_layer
            kw = dict(num_heads=H) if guided else {}
            out, _ = fn(P, pcs[lin], feats, edges, nrms[lin], *sp, drop_scale=drops.get(name), **kw)
        pre = name + '.'
        mine = [k for k in shapes if k.startswith(pre)]
        for k in mine:
            table[k].grad = None
        out.backward(up)
        bad = MF.block_mismatch(g, name, out, feats, skip, ((k[len(pre):], table[k].grad) for k in mine))
        assert not bad, (name, bad[:6])
