"""GPU: the host-side layer mirror (pcf_layers, HIP kernels underneath) against the golden vectors
the REFERENCE layers produced (outputs, input gradients, every parameter gradient), and against the
oracle on a second seed.  fp32, tolerance 1e-3 as BASELINE.json states; tighter where possible."""
import pytest
import torch

from conftest import load_golden, split

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-3, atol=1e-3)


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def cfg(**kw):
    c = Cfg(attention_type='subtraction', BATCH_NORM=True, drop_path_rate=0., dropout_rate=0., USE_VI=True,
            USE_PE=False, PCONV_OPT=False, USE_CUDA_KERNEL=True, layer_norm_guidance=False)
    c.update(kw)
    return c


def _remap_for_pconv_opt(sd):
    """PCONV_OPT=True stores the same tensors under other names (layers.py:591-602)."""
    out = {}
    for k, v in sd.items():
        k = k.replace('linear.c.', 'pconv_linear_opt.linear.') if k.startswith('linear.c.') else k
        k = k.replace('linear.bn.', 'bn.') if k.startswith('linear.bn.') else k
        if k in ('linear.weight', 'linear.bias'):
            k = 'pconv_linear_opt.' + k
        out[k] = v
    return out


def _unmap(name):
    if name.startswith('pconv_linear_opt.linear.'):
        return [name.replace('pconv_linear_opt.linear.', 'linear.c.'), name.replace('pconv_linear_opt.', '')]
    if name.startswith('bn.'):
        return ['linear.' + name]
    return [name]


def _run(layer, g, device, feat_keys, order, opt=False):
    sd = split(g, 'sd.')
    layer.load_state_dict(_remap_for_pconv_opt(sd) if opt else sd, strict=True)
    layer.to(device).train()
    a = split(g, 'in.')
    args = {}
    for k in order:
        v = a.get(k)
        if v is None:
            args[k] = None
            continue
        v = v.to(device)
        if k in feat_keys:
            v.requires_grad_(True)
        args[k] = v
    out, wn = layer(**args)
    torch.testing.assert_close(wn.cpu(), g['out.wn_in'], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.cpu(), g['out.new_feat'], **TOL)
    out.backward(g['gup'].to(device))
    for k in feat_keys:
        torch.testing.assert_close(args[k].grad.cpu(), g['gin.' + k], **TOL)
    want = split(g, 'gsd.')
    seen = 0
    for name, p in layer.named_parameters():
        keys = [k for k in _unmap(name) if k in want] if opt else [name]
        assert keys and p.grad is not None, name
        ref = want[keys[0]]
        tol = TOL
        if name.endswith('c.bias') or (opt and name == 'pconv_linear_opt.linear.bias' and 'bn.weight' in dict(layer.named_parameters())):
            tol = dict(rtol=0, atol=5e-3)     # analytically zero gradient in front of a batch-stat BN
        torch.testing.assert_close(p.grad.cpu(), ref, **tol, msg=lambda m, n=name: f'{n}: {m}')
        seen += 1
    assert seen == len(want)
    return layer


PCF_ORDER = ['dense_xyz', 'dense_feats', 'nei_inds', 'dense_xyz_norm', 'sparse_xyz', 'sparse_xyz_norm']


# edge graph: fused forward + fused three-pass backward (default) / fused forward + layer-at-a-time backward /
# every layer through its own kernels / fused edge graph but the point-level Linear_BN layers one by one / the point-level
# layers as fused contraction chains (csrc/fused_linear.hip) whatever the row count (the default uses them from 65536 rows, the
# row chains of csrc/point_chain.hip at the BASELINE widths, the layer-by-layer kernels otherwise)
CHAIN_MODES = {'fused': {}, 'layerwise_bwd': dict(EDGE_CHAIN_LAYERWISE_BACKWARD=True), 'off': dict(NO_EDGE_CHAIN=True),
               'point_layers_one_by_one': dict(NO_POINT_CHAIN=True), 'flin_point_chains': dict(FLIN_POINT_CHAINS=True)}


@pytest.mark.parametrize('mode', list(CHAIN_MODES))
@pytest.mark.parametrize('name,ci,co,cm,heads', [('pcf_self_64', 64, 64, 16, 8), ('pcf_self_32_64', 32, 64, 16, 8),
                                                 ('pcf_strided', 32, 64, 4, 4)])
def test_pcf_layer_matches_reference(device, name, ci, co, cm, heads, mode):
    import pcf_layers
    g = load_golden(name)
    layer = pcf_layers.PCFLayer(ci, co, cfg(**CHAIN_MODES[mode]), weightnet=[12, cm], num_heads=heads,
                                guidance_feat_len=32)
    _run(layer, g, device, ['dense_feats'], PCF_ORDER)


@pytest.mark.parametrize('name,ci,co,cm,heads,extra', [
    ('pcf_qk_self', 64, 64, 16, 8, dict(attention_type='qk')), ('pcf_qk_strided', 32, 64, 4, 4, dict(attention_type='qk')),
    ('pcf_ln_self', 64, 64, 16, 8, dict(layer_norm_guidance=True)), ('pcf_ln_strided', 32, 64, 4, 4, dict(layer_norm_guidance=True))])
def test_pcf_layer_guidance_ablations_match_reference(device, name, ci, co, cm, heads, extra):
    """The guidance ablations -- inner-product form (cfg.attention_type != 'subtraction' -> MultiHeadGuidanceQK,
    layers.py:77-114) and LayerNorm on query / key (cfg.layer_norm_guidance, :33-36) -- against the reference's PyTorch
    path: reference state_dict loaded strictly; output, feature gradient, every parameter gradient."""
    import pcf_layers
    g = load_golden(name)
    layer = pcf_layers.PCFLayer(ci, co, cfg(**extra), weightnet=[12, cm], num_heads=heads, guidance_feat_len=32)
    if 'attention_type' in extra:
        assert isinstance(layer.guidance_weight, pcf_layers.MultiHeadGuidanceQK)
    else:
        assert layer.guidance_weight.layer_norm
    _run(layer, g, device, ['dense_feats'], PCF_ORDER)


@pytest.mark.parametrize('name,ci,co', [('ptl_self', 32, 32), ('ptl_strided', 32, 64)])
def test_point_transformer_layer_matches_reference(device, name, ci, co):
    """PointTransformerLayer (the transformer_type ablation, layers.py:419-539) against the reference's PyTorch path."""
    import pcf_layers
    g = load_golden(name)
    layer = pcf_layers.PointTransformerLayer(ci, co, 8)
    layer.load_state_dict(split(g, 'sd.'), strict=True)
    layer.to(device).train()
    a = split(g, 'in.')
    feats = a['feats'].to(device).requires_grad_(True)
    sx = a['sparse_xyz'].to(device) if 'sparse_xyz' in a else None
    out = layer(a['xyz'].to(device), feats, a['nei_ind'].to(device), sx)
    torch.testing.assert_close(out.cpu(), g['out.new_feat'], **TOL)
    out.backward(g['gup'].to(device))
    torch.testing.assert_close(feats.grad.cpu(), g['gin.feats'], **TOL)
    want = split(g, 'gsd.')
    for pname, p in layer.named_parameters():
        assert p.grad is not None and pname in want, pname
        tol = dict(rtol=0, atol=5e-3) if pname.endswith('c.bias') else TOL     # zero gradient in front of a batch-stat BN
        torch.testing.assert_close(p.grad.cpu(), want[pname], **tol, msg=lambda m, n=pname: f'{n}: {m}')
    # the statistics the three BatchNorms tracked
    for k, v in layer.state_dict().items():
        if 'num_batches_tracked' in k:
            assert int(v) == 1, k


def test_backbone_builds_point_transformer_blocks(device):
    """cfg.transformer_type != 'PCF' puts PointTransformerLayer blocks in the guided levels (model_architecture.py:138-176)
    and the model runs forward + backward through them."""
    import pcf_layers
    import pcf_model
    c = pcf_model.Config(dict(BATCH_NORM=True, USE_XYZ=True, USE_PE=True, point_dim=3, num_level=3, grid_size=[0.1, 0.2, 0.4],
                              base_dim=16, feat_dim=[16, 32, 48], mid_dim=[4, 4, 4], mid_dim_back=1, guided_level=0, num_heads=4,
                              resblocks=[0, 1, 1], resblocks_back=[0, 0, 0], K_self=[8] * 3, K_forward=[8] * 3, K_propagate=[8] * 3,
                              num_classes=5, drop_path_rate=0., dropout_rate=0., dropout_fc=0., layer_norm_guidance=False,
                              transformer_type='PointTransformer'))
    pcf_model.get_default_configs(c, num_level=3, base_dim=16)
    c.PCONV_OPT, c.USE_CUDA_KERNEL = True, True
    torch.manual_seed(0)
    net = pcf_model.PointConvFormer_Segmentation(c).to(device).train()
    assert all(isinstance(m, pcf_layers.PointTransformerLayer) for m in net.pcf_backbone.pointconv)
    import knn_post_dataloader_utils as U
    g = torch.Generator().manual_seed(3)
    xyz = torch.rand(1500, 3, generator=g).to(device)
    nrm = torch.nn.functional.normalize(torch.randn(1500, 3, generator=g), dim=-1).to(device)
    pcs, nrms, stored = U.subsample_packed(xyz, nrm, [900, 600], [0.05, 0.12, 0.3])
    es, ef, ep = U.prepare(*U.compute_knn_packed(pcs, stored, c.K_self, c.K_forward, c.K_propagate))
    inv = U.compute_knn_inverse(pcs, es, ef, ep)
    feats = torch.randn(1, pcs[0].shape[1], 3, generator=g).to(device)
    out = net(feats, pcs, es, ef, ep, nrms, *inv)
    assert out.shape == (1, pcs[0].shape[1], 5) and torch.isfinite(out).all()
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


# (4, 1001, 4): 4004 edges per cloud, so batch boundaries fall inside 16-edge tiles
@pytest.mark.parametrize('B,N,K,cm,heads,gfl', [(2, 3000, 16, 16, 8, 32), (1, 4096, 8, 8, 4, 16), (3, 1000, 4, 16, 8, 20),
                                                (4, 1001, 4, 4, 2, 8), (1, 64, 2, 16, 8, 32)])
def test_fused_edge_chain_backward_against_layerwise(device, B, N, K, cm, heads, gfl):
    """The fused recompute backward (edge_chain_bwd.hip) against the layer-at-a-time kernels on a random
    layer: same forward, gradients of the features, of the per-point guidance term and of all 24 parameters."""
    import pcf_layers
    torch.manual_seed(5)
    xyz = torch.rand(B, N, 3, device=device)
    nrm = torch.nn.functional.normalize(torch.randn(B, N, 3, device=device), dim=-1)
    d = torch.cdist(xyz, xyz)
    nei = d.topk(K, dim=-1, largest=False).indices.contiguous()
    feats = torch.randn(B, N, 32, device=device)
    up = torch.randn(B, N, 64, device=device)
    grads = {}
    for mode in ('fused', 'layerwise_bwd'):
        torch.manual_seed(11)
        layer = pcf_layers.PCFLayer(32, 64, cfg(**CHAIN_MODES[mode]), weightnet=[12, cm], num_heads=heads,
                                    guidance_feat_len=gfl).to(device).train()
        with torch.no_grad():       # non-trivial BatchNorm affine parameters
            for m in layer.modules():
                if isinstance(m, torch.nn.BatchNorm1d) or isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.uniform_(-0.3, 0.3)
        assert layer._chain_layers(torch.empty(B, N, K, 12), nei) is not None
        x = feats.clone().requires_grad_(True)
        out, _ = layer(xyz, x, nei, nrm)
        out.backward(up)
        grads[mode] = dict(out=out.detach(), x=x.grad, **{n: p.grad for n, p in layer.named_parameters()})
    # Some gradients are analytically zero (a bias in front of a batch-stat BN; any constant shift of the guidance
    # features, which q - key cancels) and hold only rounding noise: absolute slack tied to the largest gradient.
    top = max(float(t.abs().max()) for k, t in grads['layerwise_bwd'].items() if k not in ('out', 'x'))
    for k, v in grads['fused'].items():
        ref = grads['layerwise_bwd'][k]
        scale = float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(v, ref, rtol=1e-3, atol=2e-4 * scale + 2e-5 * top, msg=lambda m, n=k: f'{n}: {m}')


@pytest.mark.parametrize('opt', [False, True])
def test_pointconv_single_matches_reference(device, opt):
    import pcf_layers
    g = load_golden('pointconv_single')
    c = cfg(BATCH_NORM=False, USE_PE=False, USE_VI=False, PCONV_OPT=opt)
    layer = pcf_layers.PointConv(3, 32, c, weightnet=[3, 16])
    _run(layer, g, device, ['dense_feats'], ['dense_xyz', 'dense_feats', 'nei_inds'], opt=opt)


@pytest.mark.parametrize('opt', [False, True])
def test_pointconv_vi_pe_matches_reference(device, opt):
    import pcf_layers
    g = load_golden('pointconv_vi_pe')
    layer = pcf_layers.PointConv(6, 64, cfg(USE_PE=True, PCONV_OPT=opt), weightnet=[12, 16])
    _run(layer, g, device, ['dense_feats'], ['dense_xyz', 'dense_feats', 'nei_inds', 'dense_xyz_norm'], opt=opt)


@pytest.mark.parametrize('opt', [False, True])
def test_stride_pe_matches_reference(device, opt):
    import pcf_layers
    g = load_golden('stride_pe')
    layer = pcf_layers.PointConvStridePE(64, 64, cfg(USE_PE=True, PCONV_OPT=opt), weightnet=[12, 16])
    _run(layer, g, device, ['dense_feats'], PCF_ORDER, opt=opt)


@pytest.mark.parametrize('opt', [False, True])
def test_transpose_pe_matches_reference(device, opt):
    import pcf_layers
    g = load_golden('transpose_pe')
    layer = pcf_layers.PointConvTransposePE(128, 64, cfg(USE_PE=True, PCONV_OPT=opt), weightnet=[12, 1], mlp2=[64, 64])
    _run(layer, g, device, ['sparse_feats', 'dense_feats'],
         ['sparse_xyz', 'sparse_feats', 'nei_inds', 'sparse_xyz_norm', 'dense_xyz', 'dense_xyz_norm', 'dense_feats'], opt=opt)


def test_edge_geometry_and_gathers_against_oracle(device):
    import pcf_fused
    from oracle import pcf_oracle as O
    g = torch.Generator().manual_seed(3)
    B, N, M, K, C = 2, 150, 60, 7, 20
    xyz, nrm = torch.rand(B, N, 3, generator=g), torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1)
    cxyz, cnrm = torch.rand(B, M, 3, generator=g), torch.nn.functional.normalize(torch.randn(B, M, 3, generator=g), dim=-1)
    idx = torch.randint(0, N, (B, M, K), generator=g)
    d = lambda t: t.to(device)
    rel, vi = pcf_fused.edge_geometry(d(xyz), d(nrm), d(idx), d(cxyz), d(cnrm))
    wrel = O.gather_rows(xyz, idx) - cxyz.unsqueeze(2)
    torch.testing.assert_close(rel.cpu(), wrel, rtol=0, atol=0)
    torch.testing.assert_close(vi.cpu(), O.vi_features(wrel, O.gather_rows(nrm, idx), cnrm), rtol=1e-4, atol=1e-5)
    vi2 = pcf_fused.vi_from_gathered(rel, d(O.gather_rows(nrm, idx)), d(cnrm))
    torch.testing.assert_close(vi2, vi, rtol=0, atol=0)
    # reference golden for the VI transform, zero offsets (self edges) included
    gg = load_golden('vi_transform')
    _, vig = pcf_fused.edge_geometry(d(gg['xyz'][None]), d(gg['nrm'][None]), d(gg['idx'][None]), d(gg['xyz'][None]),
                                     d(gg['nrm'][None]))
    torch.testing.assert_close(vig[0].cpu(), gg['vi'], rtol=1e-4, atol=1e-5)
    # differentiable gathers
    t = torch.randn(B, N, C, generator=g)
    td = d(t).requires_grad_(True)
    got = pcf_fused.gather_rows(td, d(idx))
    torch.testing.assert_close(got.cpu(), O.gather_rows(t, idx), rtol=0, atol=0)
    up = torch.randn(got.shape, generator=g)
    got.backward(d(up))
    tr = t.clone().requires_grad_(True)
    O.gather_rows(tr, idx).backward(up)
    torch.testing.assert_close(td.grad.cpu(), tr.grad, rtol=1e-5, atol=1e-5)
    td.grad = None
    mx = pcf_fused.gather_max(td, d(idx))
    want = O.gather_rows(tr, idx).max(2)[0]
    torch.testing.assert_close(mx.cpu(), want, rtol=0, atol=0)
    up = torch.randn(mx.shape, generator=g)
    mx.backward(d(up))
    tr.grad = None
    want.backward(up)
    torch.testing.assert_close(td.grad.cpu(), tr.grad, rtol=1e-5, atol=1e-5)


def test_linear_bn_fuse(device):
    """Inference-time folding of BN into the linear (layer_utils.py:260-270)."""
    import pcf_layers
    torch.manual_seed(0)
    m = pcf_layers.Linear_BN(12, 8).to(device)
    x = torch.randn(2, 50, 16, 12, device=device)
    m.train()
    for _ in range(3):
        m(x)
    m.eval()
    torch.testing.assert_close(m.fuse()(x), m(x), rtol=1e-4, atol=1e-5)


ROWLIN_CASES = [
    # R-shape, Cin, Cout, act, bn, training
    ((2, 300, 16), 12, 32, 1, True, True),
    ((1, 500, 16), 64, 8, 1, True, True),
    ((1, 400, 16), 8, 8, 3, True, True),
    ((1, 200, 7), 8, 16, 1, True, True),
    ((1, 777), 64, 16, 2, True, True),
    ((1, 1000), 16, 32, 0, True, True),
    ((1, 100, 16), 3, 16, 1, True, True),
    ((1, 90, 16), 16, 16, 1, True, False),        # running statistics
    ((1, 50, 16), 64, 8, 1, False, True),         # no BN (cfg.BATCH_NORM False)
    ((1, 64, 12), 48, 32, 1, True, True),
    ((1, 33, 5), 5, 1, 1, True, True),            # odd sizes: scalar row access, Cout = 1 (mid_dim_back)
    ((1, 120, 16), 64, 64, 2, True, True),
]


@pytest.mark.parametrize('shape,cin,cout,act,bn,training', ROWLIN_CASES)
def test_linear_bn_act_against_torch(device, shape, cin, cout, act, bn, training):
    """Fused Linear+BatchNorm+activation (csrc/edge_mlp.hip) vs the same three torch modules on CPU
    in fp64: output, input gradient, every parameter gradient and the running statistics."""
    import pcf_fused
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(*shape, cin, generator=g) + 0.3
    lin = torch.nn.Linear(cin, cout)
    bnm = torch.nn.BatchNorm1d(cout) if bn else None
    if bn:
        with torch.no_grad():
            bnm.weight.copy_(torch.rand(cout, generator=g) + 0.5)
            bnm.bias.copy_(torch.randn(cout, generator=g) * 0.2)
            bnm.running_mean.copy_(torch.randn(cout, generator=g) * 0.1)
            bnm.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    import copy
    lin_d, bn_d = copy.deepcopy(lin).to(device), (copy.deepcopy(bnm).to(device) if bn else None)
    ref_lin, ref_bn = copy.deepcopy(lin).double(), (copy.deepcopy(bnm).double() if bn else None)
    for m in (bn_d, ref_bn):
        if m is not None:
            m.train(training)
    actf = {0: lambda t: t, 1: torch.relu, 2: lambda t: torch.nn.functional.leaky_relu(t, 0.1), 3: torch.sigmoid}[act]
    xr = x.double().requires_grad_(True)
    z = ref_lin(xr)
    if bn:
        z = ref_bn(z.reshape(-1, cout)).view(z.shape)
    want = actf(z)
    up = torch.randn(want.shape, generator=g)
    want.backward(up.double())
    xd = x.to(device).requires_grad_(True)
    got = pcf_fused.linear_bn_act(xd, lin_d.weight, lin_d.bias, bn_d, act, training)
    got.backward(up.to(device))
    tol = dict(rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(got.cpu(), want.float(), **tol)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad.float(), **tol)
    gscale = max(1.0, float(ref_lin.weight.grad.abs().max()))
    torch.testing.assert_close(lin_d.weight.grad.cpu(), ref_lin.weight.grad.float(), rtol=2e-4, atol=2e-4 * gscale)
    torch.testing.assert_close(lin_d.bias.grad.cpu(), ref_lin.bias.grad.float(), rtol=2e-4, atol=2e-3 * gscale)
    if bn:
        torch.testing.assert_close(bn_d.weight.grad.cpu(), ref_bn.weight.grad.float(), rtol=2e-4, atol=2e-4 * gscale)
        torch.testing.assert_close(bn_d.bias.grad.cpu(), ref_bn.bias.grad.float(), rtol=2e-4, atol=2e-4 * gscale)
        torch.testing.assert_close(bn_d.running_mean.cpu(), ref_bn.running_mean.float(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(bn_d.running_var.cpu(), ref_bn.running_var.float(), rtol=1e-5, atol=1e-6)
        assert int(bn_d.num_batches_tracked) == int(ref_bn.num_batches_tracked)


@pytest.mark.parametrize('use_max', [False, True])
def test_guidance_diff_against_torch(device, use_max):
    import pcf_fused
    from oracle import pcf_oracle as O
    g = torch.Generator().manual_seed(5)
    B, N, M, K, G, P = 2, 90, 40, 6, 32, 32
    gx = torch.randn(B, N, G, generator=g)
    pe = torch.randn(B, M, K, P, generator=g)
    idx = torch.randint(0, N, (B, M, K), generator=g)
    gr, pr = gx.clone().requires_grad_(True), pe.clone().requires_grad_(True)
    q = torch.cat([O.gather_rows(gr, idx), pr], -1)
    key = q.max(2, keepdim=True)[0] if use_max else q[:, :, :1]
    want = q - key
    up = torch.randn(want.shape, generator=g)
    want.backward(up)
    gd, pd = gx.to(device).requires_grad_(True), pe.to(device).requires_grad_(True)
    got = pcf_fused.guidance_diff(gd, idx.to(device), pd, use_max)
    got.backward(up.to(device))
    torch.testing.assert_close(got.cpu(), want, rtol=0, atol=0)
    torch.testing.assert_close(pd.grad.cpu(), pr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gd.grad.cpu(), gr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('K,bn', [(16, True), (8, True), (4, False), (64, True)])
def test_split_guidance_first_layer(device, K, bn):
    """linear_bn_act with the gathered per-point term and key subtraction == Linear_BN(q - key) formed
    explicitly (layers.py:372-382), outputs and all gradients, vs torch fp64 on the CPU."""
    import pcf_fused
    from oracle import pcf_oracle as O
    g = torch.Generator().manual_seed(K)
    B, N, G, P, Cout = 2, 70, 32, 32, 8
    M = N
    gx = torch.randn(B, N, G, generator=g)
    pe = torch.randn(B, M, K, P, generator=g)
    idx = torch.randint(0, N, (B, M, K), generator=g)
    W = torch.randn(Cout, G + P, generator=g) / 8
    b = torch.randn(Cout, generator=g)
    gam, bet = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    # reference in fp64
    gr, pr, Wr, br = (t.double().requires_grad_(True) for t in (gx, pe, W, b))
    gamr, betr = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    q = torch.cat([O.gather_rows(gr, idx), pr], -1)
    z = torch.nn.functional.linear(q - q[:, :, :1], Wr, br)
    if bn:
        flat = z.reshape(-1, Cout)
        z = ((flat - flat.mean(0)) / torch.sqrt(flat.var(0, unbiased=False) + 1e-5) * gamr + betr).view(z.shape)
    want = torch.relu(z)
    up = torch.randn(want.shape, generator=g)
    want.backward(up.double())
    # HIP
    d = lambda t: t.to(device).requires_grad_(True)
    gd, pd, Wd, bd = d(gx), d(pe), d(W), d(b)
    bnm = None
    if bn:
        bnm = torch.nn.BatchNorm1d(Cout).to(device).train()
        with torch.no_grad():
            bnm.weight.copy_(gam); bnm.bias.copy_(bet)
    u = pcf_fused.linear_bn_act(gd, Wd[:, :G], torch.zeros(Cout, device=device), None, 0, True)
    got = pcf_fused.linear_bn_act(pd, Wd[:, G:], bd, bnm, 1, True, gadd=u, gidx=idx.to(device), group=K)
    got.backward(up.to(device))
    tol = dict(rtol=3e-4, atol=3e-4)
    torch.testing.assert_close(got.cpu(), want.float(), **tol)
    torch.testing.assert_close(pd.grad.cpu(), pr.grad.float(), **tol)
    torch.testing.assert_close(gd.grad.cpu(), gr.grad.float(), **tol)
    sc = max(1.0, float(Wr.grad.abs().max()))
    torch.testing.assert_close(Wd.grad.cpu(), Wr.grad.float(), rtol=3e-4, atol=3e-4 * sc)
    torch.testing.assert_close(bd.grad.cpu(), br.grad.float(), rtol=3e-4, atol=3e-3 * sc)
    if bn:
        torch.testing.assert_close(bnm.weight.grad.cpu(), gamr.grad.float(), rtol=3e-4, atol=3e-4 * sc)
        torch.testing.assert_close(bnm.bias.grad.cpu(), betr.grad.float(), rtol=3e-4, atol=3e-4 * sc)


def _model_cfg(opt):
    import pcf_model
    c = pcf_model.Config(USE_PE=True, num_classes=5, PCONV_OPT=opt, USE_CUDA_KERNEL=True)
    pcf_model.get_default_configs(c, num_level=3, base_dim=16)
    c.update(feat_dim=[16, 32, 48], mid_dim=[4, 4, 4], mid_dim_back=1, guided_level=0, num_heads=4,
             resblocks=[0, 2, 1], resblocks_back=[0, 0, 0])
    return c


def _opt_key_map(net):
    """my PCONV_OPT=True parameter / buffer names -> the names the reference uses with PCONV_OPT=False
    (layers.py:591-602: `pconv_linear_opt.linear` + `bn` versus `linear.c` + `linear.bn`)."""
    m = {}
    for name, mod in net.named_modules():
        if hasattr(mod, 'pconv_linear_opt'):
            pre = name + '.'
            for t in ('weight', 'bias'):
                m[pre + 'pconv_linear_opt.linear.' + t] = pre + 'linear.c.' + t
            if hasattr(mod, 'bn'):
                for t in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
                    m[pre + 'bn.' + t] = pre + 'linear.bn.' + t
    return m


@pytest.mark.parametrize('opt', [False, True])
def test_segmentation_model_matches_reference(device, opt):
    """The whole PointConvFormer_Segmentation graph (PointConv + StridePE level 0, strided and plain
    PCFLayers, TransposePE decoder, head) against the reference model's golden: logits, feature gradient
    and all parameter gradients; PCONV_OPT=True additionally routes the PointConv family through the
    fused aggregate+linear op with the CSR backward."""
    import pcf_model
    import knn_post_dataloader_utils as U
    g = load_golden('model_seg3')
    net = pcf_model.PointConvFormer_Segmentation(_model_cfg(opt))
    kmap = _opt_key_map(net) if opt else {}
    ref_sd = split(g, 'sd.')
    net.load_state_dict({k: ref_sd[kmap.get(k, k)] for k in net.state_dict()}, strict=True)
    net.to(device).train()
    L = 3
    pcs = [g[f'in.xyz{l}'][None].to(device) for l in range(L)]
    nrms = [g[f'in.nrm{l}'][None].to(device) for l in range(L)]
    es = [g[f'in.edges_self{l}'].to(device) for l in range(L)]
    ef = [g[f'in.edges_forward{l}'].to(device) for l in range(L - 1)]
    ep = [g[f'in.edges_propagate{l}'].to(device) for l in range(L - 1)]
    feats = g['in.features'].to(device).requires_grad_(True)
    inv = U.compute_knn_inverse(pcs, es, ef, ep) if opt else (None, None, None)
    out = net(feats, pcs, es, ef, ep, nrms, *inv)
    torch.testing.assert_close(out.cpu(), g['out'], **TOL)
    out.backward(g['gup'].to(device))
    torch.testing.assert_close(feats.grad.cpu(), g['gin.features'], rtol=2e-3, atol=2e-3)
    want = split(g, 'gsd.')
    bad = []
    for name, p in net.named_parameters():
        ref_name = kmap.get(name, name)
        ref = want[ref_name]
        zero_grad_bias = ref_name.endswith('c.bias') and ref_name[:-6] + 'bn.weight' in want
        scale = max(1.0, float(ref.abs().max()))
        if not torch.allclose(p.grad.cpu(), ref, rtol=2e-3, atol=(5e-3 if zero_grad_bias else 2e-3) * scale):
            bad.append((name, float((p.grad.cpu() - ref).abs().max()), scale))
    assert not bad, bad[:8]


@pytest.mark.parametrize('rows,cin,cout,act,bn,training', [
    (5000, 256, 32, 1, True, True), (777, 416, 256, 1, True, True), (1200, 96, 192, 2, True, True),
    (300, 1536, 192, 0, True, False), (640, 128, 20, 0, False, True), (1000, 70, 70, 3, True, True)])
def test_wide_linear_bn_act_against_torch(device, rows, cin, cout, act, bn, training):
    """MFMA contraction + column-wise BN kernels vs torch fp64 on the CPU: output, dx, dW, db, dgamma, dbeta,
    running statistics."""
    import copy
    import pcf_fused
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(1, rows, cin, generator=g) + 0.2
    lin = torch.nn.Linear(cin, cout)
    bnm = torch.nn.BatchNorm1d(cout) if bn else None
    if bn:
        with torch.no_grad():
            bnm.weight.copy_(torch.rand(cout, generator=g) + 0.5)
            bnm.bias.copy_(torch.randn(cout, generator=g) * 0.2)
            bnm.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    lin_d, bn_d = copy.deepcopy(lin).to(device), (copy.deepcopy(bnm).to(device).train(training) if bn else None)
    ref_lin, ref_bn = copy.deepcopy(lin).double(), (copy.deepcopy(bnm).double().train(training) if bn else None)
    actf = {0: lambda t: t, 1: torch.relu, 2: lambda t: torch.nn.functional.leaky_relu(t, 0.1), 3: torch.sigmoid}[act]
    xr = x.double().requires_grad_(True)
    zz = ref_lin(xr)
    if bn:
        zz = ref_bn(zz.reshape(-1, cout)).view(zz.shape)
    want = actf(zz)
    up = torch.randn(want.shape, generator=g)
    want.backward(up.double())
    xd = x.to(device).requires_grad_(True)
    got = pcf_fused.wide_linear_bn_act(xd, lin_d.weight, lin_d.bias, bn_d, act, training)
    got.backward(up.to(device))
    tol = dict(rtol=3e-4, atol=3e-4)
    torch.testing.assert_close(got.cpu(), want.float(), **tol)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad.float(), **tol)
    sc = max(1.0, float(ref_lin.weight.grad.abs().max()))
    torch.testing.assert_close(lin_d.weight.grad.cpu(), ref_lin.weight.grad.float(), rtol=3e-4, atol=3e-4 * sc)
    torch.testing.assert_close(lin_d.bias.grad.cpu(), ref_lin.bias.grad.float(), rtol=3e-4, atol=3e-3 * sc)
    if bn:
        torch.testing.assert_close(bn_d.weight.grad.cpu(), ref_bn.weight.grad.float(), rtol=3e-4, atol=3e-4 * sc)
        torch.testing.assert_close(bn_d.bias.grad.cpu(), ref_bn.bias.grad.float(), rtol=3e-4, atol=3e-4 * sc)
        torch.testing.assert_close(bn_d.running_mean.cpu(), ref_bn.running_mean.float(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(bn_d.running_var.cpu(), ref_bn.running_var.float(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('rows,C,act,training', [(1408, 512, 1, True), (40, 320, 1, True), (30000, 64, 1, True),
                                                 (5000, 192, 2, True), (777, 70, 1, False), (17, 3, 0, True)])
def test_bn_act_against_torch(device, rows, C, act, training):
    """Stand-alone act(BatchNorm1d(z)) (the BN + ReLU behind the fused aggregate + linear, layers.py:709/721) vs torch
    fp64: output, dz, dgamma, dbeta, running statistics, batch counter.  The shapes cover short-and-wide matrices
    (row ranges x column chunks of the reduction grid), columns that are not a multiple of 4, and inference."""
    import copy
    import pcf_fused
    g = torch.Generator().manual_seed(rows + C)
    z = torch.randn(2, rows, C, generator=g) * (torch.rand(C, generator=g) + 0.5) + torch.randn(C, generator=g)
    bnm = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bnm.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bnm.bias.copy_(torch.randn(C, generator=g) * 0.2)
        bnm.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn_d, ref_bn = copy.deepcopy(bnm).to(device).train(training), copy.deepcopy(bnm).double().train(training)
    actf = {0: lambda t: t, 1: torch.relu, 2: lambda t: torch.nn.functional.leaky_relu(t, 0.1)}[act]
    zr = z.double().requires_grad_(True)
    want = actf(ref_bn(zr.reshape(-1, C)).view(zr.shape))
    up = torch.randn(want.shape, generator=g)
    want.backward(up.double())
    zd = z.to(device).requires_grad_(True)
    got = pcf_fused.bn_act(zd, bn_d, act, training)
    got.backward(up.to(device))
    tol = dict(rtol=3e-4, atol=3e-4)
    torch.testing.assert_close(got.cpu(), want.float(), **tol)
    torch.testing.assert_close(zd.grad.cpu(), zr.grad.float(), **tol)
    sc = max(1.0, float(ref_bn.weight.grad.abs().max()))
    torch.testing.assert_close(bn_d.weight.grad.cpu(), ref_bn.weight.grad.float(), rtol=3e-4, atol=3e-4 * sc)
    torch.testing.assert_close(bn_d.bias.grad.cpu(), ref_bn.bias.grad.float(), rtol=3e-4, atol=3e-4 * sc)
    torch.testing.assert_close(bn_d.running_mean.cpu(), ref_bn.running_mean.float(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn_d.running_var.cpu(), ref_bn.running_var.float(), rtol=1e-5, atol=1e-6)
    assert int(bn_d.num_batches_tracked) == int(ref_bn.num_batches_tracked)


@pytest.mark.parametrize('rows,cin,cout,act,bn,training', [(5000, 32, 64, 2, True, True), (999, 128, 256, 2, True, True),
                                                          (640, 96, 70, 1, False, True), (300, 160, 64, 2, True, False)])
def test_wide_linear_bn_act_with_residual(device, rows, cin, cout, act, bn, training):
    """y = act(BN(x W^T + b) + residual) in the column-BN kernels (the tail of every residual block) vs torch fp64:
    output and the gradients of x, the residual and all parameters."""
    import copy
    import pcf_fused
    g = torch.Generator().manual_seed(rows + cout)
    x = torch.randn(1, rows, cin, generator=g) + 0.2
    res = torch.randn(1, rows, cout, generator=g)
    lin = torch.nn.Linear(cin, cout)
    bnm = torch.nn.BatchNorm1d(cout) if bn else None
    if bn:
        with torch.no_grad():
            bnm.weight.copy_(torch.rand(cout, generator=g) + 0.5)
            bnm.bias.copy_(torch.randn(cout, generator=g) * 0.2)
            bnm.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    lin_d, bn_d = copy.deepcopy(lin).to(device), (copy.deepcopy(bnm).to(device).train(training) if bn else None)
    ref_lin, ref_bn = copy.deepcopy(lin).double(), (copy.deepcopy(bnm).double().train(training) if bn else None)
    actf = {1: torch.relu, 2: lambda t: torch.nn.functional.leaky_relu(t, 0.1)}[act]
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    zz = ref_lin(xr)
    if bn:
        zz = ref_bn(zz.reshape(-1, cout)).view(zz.shape)
    want = actf(zz + rr)
    up = torch.randn(want.shape, generator=g)
    want.backward(up.double())
    xd, rd = x.to(device).requires_grad_(True), res.to(device).requires_grad_(True)
    got = pcf_fused.wide_linear_bn_act(xd, lin_d.weight, lin_d.bias, bn_d, act, training, residual=rd)
    got.backward(up.to(device))
    tol = dict(rtol=3e-4, atol=3e-4)
    torch.testing.assert_close(got.cpu(), want.float(), **tol)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad.float(), **tol)
    torch.testing.assert_close(rd.grad.cpu(), rr.grad.float(), **tol)
    sc = max(1.0, float(ref_lin.weight.grad.abs().max()))
    torch.testing.assert_close(lin_d.weight.grad.cpu(), ref_lin.weight.grad.float(), rtol=3e-4, atol=3e-4 * sc)
    if bn:
        torch.testing.assert_close(bn_d.weight.grad.cpu(), ref_bn.weight.grad.float(), rtol=3e-4, atol=3e-4 * sc)
        torch.testing.assert_close(bn_d.bias.grad.cpu(), ref_bn.bias.grad.float(), rtol=3e-4, atol=3e-4 * sc)


@pytest.mark.parametrize('B,M,K', [(1, 1000, 16), (2, 37, 5), (1, 4, 16), (3, 211, 16)])
def test_edge_geometry_store_paths(device, B, M, K):
    """The VI rows leave the kernel as contiguous KiB per full wave and row by row in the ragged last wave: both paths,
    edge counts that are and are not multiples of 64, against the oracle."""
    import pcf_fused
    from oracle import pcf_oracle as O
    g = torch.Generator().manual_seed(B * 1000 + M)
    N = max(M, K) + 5
    xyz, nrm = torch.rand(B, N, 3, generator=g), torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1)
    cxyz, cnrm = torch.rand(B, M, 3, generator=g), torch.nn.functional.normalize(torch.randn(B, M, 3, generator=g), dim=-1)
    idx = torch.randint(0, N, (B, M, K), generator=g)
    d = lambda t: t.to(device)
    for want_rel in (True, False):
        rel, vi = pcf_fused.edge_geometry(d(xyz), d(nrm), d(idx), d(cxyz), d(cnrm), want_rel=want_rel)
        wrel = O.gather_rows(xyz, idx) - cxyz.unsqueeze(2)
        assert (rel is None) == (not want_rel)
        if want_rel:
            torch.testing.assert_close(rel.cpu(), wrel, rtol=0, atol=0)
        torch.testing.assert_close(vi.cpu(), O.vi_features(wrel, O.gather_rows(nrm, idx), cnrm), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('rows,cin,cout,training', [((2, 500, 16), 12, 16, True), ((1, 333, 16), 3, 16, True),
                                                    ((3, 64, 8), 12, 1, True), ((1, 4096, 16), 12, 4, True),
                                                    ((2, 100, 8), 3, 16, False)])
def test_weightnet_chain_against_layer_at_a_time(device, rows, cin, cout, training):
    """WeightNet (cin -> 8 -> 8 -> C_mid) through the fused chain (four passes forward, three backward) against the
    same module run layer by layer: output, every parameter gradient, running statistics.  Includes the decoder's
    C_mid = 1 and the 3-channel (no VI) input."""
    import copy
    import pcf_layers
    torch.manual_seed(rows[1] + cin)
    wn = pcf_layers.WeightNet(cin, cout, efficient=True).to(device)
    with torch.no_grad():
        for m in wn.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(wn)
    ref.no_chain = True
    wn.train(training)
    ref.train(training)
    x = torch.randn(*rows, cin, device=device)
    import pcf_fused
    assert pcf_fused.weightnet_chain_supported(cin, (8, 8), cout, x.numel() // cin)
    out = wn(x)
    want = ref(x)
    torch.testing.assert_close(out, want, rtol=1e-4, atol=1e-5)
    if not training:
        return
    up = torch.randn_like(out)
    out.backward(up)
    want.backward(up)
    top = max(float(p.grad.abs().max()) for p in ref.parameters())
    for (n, p), (_, q) in zip(wn.named_parameters(), ref.named_parameters()):
        scale = float(q.grad.abs().max()) + 1e-12
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-3, atol=2e-4 * scale + 2e-5 * top, msg=lambda m, k=n: f'{k}: {m}')
    for (n, b), (_, c) in zip(wn.named_buffers(), ref.named_buffers()):
        torch.testing.assert_close(b.float(), c.float(), rtol=1e-4, atol=1e-5, msg=lambda m, k=n: f'{k}: {m}')


@pytest.mark.parametrize('rows,hidden,cout', [((1, 333, 16), 16, 16), ((2, 500, 16), 32, 32), ((1, 77, 3), 64, 32),
                                              ((1, 80000, 16), 32, 32), ((1, 1000, 16), 128, 32)])
def test_pe_chain_against_float64_and_layer_at_a_time(device, rows, hidden, cout):
    """pe_convs = WeightNet(3, cout, [hidden]) (layers.py:599-604, 941-946) through the row chain over the edges (three
    passes forward, three backward, nothing but the output stored): output, running statistics and the eight parameter
    gradients against float64 autograd at 1e-3 of the scale, and against the same module run layer by layer.  hidden = 128
    (the 512-wide levels) is not instantiated and must take the layer-by-layer path."""
    import copy
    import pcf_layers
    import pcf_fused
    import torch.nn.functional as F
    torch.manual_seed(rows[1] + hidden)
    wn = pcf_layers.WeightNet(3, cout, hidden_unit=[hidden], efficient=True).to(device)
    with torch.no_grad():
        for m in wn.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(wn)
    ref.no_chain = True
    wn.train()
    ref.train()
    assert pcf_fused.pe_chain_supported(3, (hidden,), cout) == (hidden != 128)
    rel = torch.randn(*rows, 3, device=device) * 0.3
    up = torch.randn(*rows, cout, device=device)
    pcf_cuda = __import__('pcf_cuda')
    pcf_cuda.launch_log(True)
    out = wn(rel)
    out.backward(up)
    names = pcf_cuda.read_launch_log()
    pcf_cuda.launch_log(False)
    assert any('pe_chain' in n for n in names) == (hidden != 128), names
    want = ref(rel)
    want.backward(up)
    # float64 reference
    P = {n: p.detach().double().cpu().requires_grad_(True) for n, p in ref.named_parameters()}
    x = rel.double().cpu().reshape(-1, 3)

    def bn(z, g, b):
        return (z - z.mean(0)) * torch.rsqrt(z.var(0, unbiased=False) + 1e-5) * g + b

    z1 = x @ P['mlp_convs.0.c.weight'].t() + P['mlp_convs.0.c.bias']
    y1 = F.relu(bn(z1, P['mlp_convs.0.bn.weight'], P['mlp_convs.0.bn.bias']))
    z2 = y1 @ P['mlp_convs.1.c.weight'].t() + P['mlp_convs.1.c.bias']
    o64 = F.relu(bn(z2, P['mlp_convs.1.bn.weight'], P['mlp_convs.1.bn.bias']))
    o64.backward(up.double().cpu().reshape(-1, cout))

    def close(got, ref_, what, tol):
        ref_ = ref_.float()
        torch.testing.assert_close(got.detach().cpu().reshape(ref_.shape), ref_, rtol=tol, atol=tol * max(1.0, float(ref_.abs().max())),
                                   msg=lambda m: f'{what}: {m}')

    close(out, o64.detach().reshape(out.shape), 'out vs float64', 1e-3)
    for n, p in wn.named_parameters():
        close(p.grad, P[n].grad, n + ' vs float64', 2e-3)
    close(wn.mlp_convs[0].bn.running_mean, 0.1 * z1.detach().mean(0), 'running_mean 1', 1e-3)
    close(wn.mlp_convs[1].bn.running_var, 0.9 + 0.1 * z2.detach().var(0, unbiased=True), 'running_var 2', 1e-3)
    close(out, want.detach().cpu(), 'out vs layer by layer', 1e-4)
    for (n, b), (_, c) in zip(wn.named_buffers(), ref.named_buffers()):
        close(b.float(), c.float().cpu(), n, 1e-4)
    assert float(pcf_fused._tickets(device).abs().sum()) == 0.0


@pytest.mark.parametrize('K,n_dense,n_out', [(16, 3000, 700), (4, 1000, 400), (8, 2000, 512)])
def test_strided_pcf_layer_chain_against_layer_by_layer(device, K, n_dense, n_out):
    """Strided PCFLayer (key = maximum of the query over the neighbourhood, layers.py:372-375) through the fused edge chain
    (maximum-key form: the gathered half of the key per centre, the positional half and its arg-max inside the kernels) against
    the same layer with every edge layer through its own kernels (`NO_EDGE_CHAIN`, itself held to the reference by the
    pcf_strided golden).  A third of the neighbour lists repeat entries, so positional maxima are attained by several edges:
    the gradient must go to the first one, as torch.max does.  Output, feature gradient and every parameter gradient."""
    import pcf_cuda
    import pcf_layers
    g = torch.Generator().manual_seed(K)
    xyz = torch.rand(1, n_dense, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(1, n_dense, 3, generator=g), dim=-1)
    ctr = torch.randperm(n_dense, generator=g)[:n_out]
    sxyz, snrm = xyz[:, ctr], nrm[:, ctr]
    idx = torch.randint(0, n_dense, (1, n_out, K), generator=g)
    idx[:, ::3, 1::2] = idx[:, ::3, 0:-1:2]                  # repeated neighbours: ties of the maximum
    feats = torch.randn(1, n_dense, 64, generator=g)
    up = torch.randn(1, n_out, 128, generator=g)
    res = {}
    for mode in ('fused', 'off'):
        torch.manual_seed(11)
        layer = pcf_layers.PCFLayer(64, 128, cfg(**CHAIN_MODES[mode]), weightnet=[12, 16], num_heads=8,
                                    guidance_feat_len=32).to(device).train()
        with torch.no_grad():
            for m in layer.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.uniform_(-0.3, 0.3)
        x = feats.clone().to(device).requires_grad_(True)
        pcf_cuda.launch_log(True)
        out, _ = layer(xyz.to(device), x, idx.to(device), nrm.to(device), sxyz.to(device), snrm.to(device))
        out.backward(up.to(device))
        names = pcf_cuda.read_launch_log()
        pcf_cuda.launch_log(False)
        assert any('guidance_diff' in n for n in names) == (mode == 'off'), names      # the explicit query - max(query) kernel
        res[mode] = dict(out=out.detach(), x=x.grad, **{n: p.grad for n, p in layer.named_parameters()},
                         **{'buf.' + n: b.clone().float() for n, b in layer.named_buffers()})
    # Forward: 1e-3 of the scale.  Backward: with 400-700 centres ONE activation whose argument rounds to opposite sides of zero
    # in the two forms (seen: a single LeakyReLU of the output, arguments 3e-6 apart) moves every gradient by percents through
    # the BatchNorms, so the gradients are held by error norms -- an adjoint bug of the maximum key is O(1) in mlp_conv / g1.
    for k, v in res['fused'].items():
        ref = res['off'][k]
        scale = float(ref.abs().max()) + 1e-12
        if k == 'out' or k.startswith('buf.'):
            torch.testing.assert_close(v, ref, rtol=1e-3, atol=1e-3 * scale, msg=lambda m, n=k: f'{n}: {m}')
        elif k == 'x':
            bad = ((v - ref).abs() > 1e-3 * scale + 1e-3 * ref.abs()).any(dim=-1)
            assert int(bad.sum()) <= max(4, n_dense // 50), f'{int(bad.sum())} rows of the feature gradient differ'
        elif float(ref.abs().max()) > 1e-3:                # biases in front of a batch-statistics BatchNorm: zero, noise
            err = float((v - ref).norm() / ref.norm())
            assert err < 5e-2, f'{k}: relative error {err:.3e}'


def test_pe_chain_is_repeatable_with_many_short_workgroups(device):
    """Regression: 51 973 x 16 edges = 2048 workgroups of six tiles per wave.  The hand-over of the per-workgroup partial sums
    to the workgroup that finishes last used "atomic store, workgroup-scope fence, ticket"; that fence compiles to no
    vector-memory wait on gfx950, the ticket overtook the store about one launch in ten, and a stale partial list entered the
    batch statistics (csrc/flin_common.h).  Twelve runs must agree bit for bit, outputs and all parameter gradients."""
    import pcf_fused as PF
    torch.manual_seed(0)
    E, H, L = 51973 * 16, 32, 32
    rel = torch.randn(E, 3, device=device) * 0.3
    base = [torch.randn(H, 3), torch.randn(H) * 0.1, torch.rand(H) + 0.5, torch.randn(H) * 0.1,
            torch.randn(L, H) / H ** 0.5, torch.randn(L) * 0.1, torch.rand(L) + 0.5, torch.randn(L) * 0.1]
    up = torch.randn(E, L, device=device)
    first = None
    for rep in range(12):
        bn1, bn2 = torch.nn.BatchNorm1d(H).to(device), torch.nn.BatchNorm1d(L).to(device)
        P = [t.clone().to(device).requires_grad_(True) for t in base]
        out = PF._PEChain.apply((bn1, bn2), rel, *P)
        out.backward(up)
        got = [out.detach(), bn1.running_mean, bn2.running_var] + [p.grad for p in P]
        if first is None:
            first = [t.clone() for t in got]
        else:
            for i, (a, b) in enumerate(zip(got, first)):
                assert torch.equal(a, b), f'run {rep}: tensor {i} differs from the first run by {float((a - b).abs().max()):.3e}'
    assert float(PF._tickets(device).abs().sum()) == 0.0


def test_fused_edge_chain_full_size(device):
    """BASELINE's full size (one packed cloud of 80 000 points, K = 16, 1.28 M edges): the fused three-pass backward
    (weight gradients of the first layers from moments accumulated over all edges) against the layer-at-a-time
    kernels behind the same fused forward.  Parameter gradients within 1e-3 of the largest entry; analytically
    zero ones (biases in front of a batch-stat BN, the guidance shift) are rounding noise on both sides."""
    import pcf_cuda
    import pcf_layers
    g = torch.Generator().manual_seed(1)
    N, K = 80000, 16
    xyz = torch.rand(1, N, 3, generator=g).to(device)
    nrm = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=g), dim=-1).to(device)
    feats = torch.randn(1, N, 64, generator=g).to(device)
    off = torch.tensor([0, N], dtype=torch.int32, device=device)
    idx = pcf_cuda.knn_packed(xyz[0], xyz[0], off, off, K)[None].contiguous()
    up = torch.randn(1, N, 64, generator=g).to(device)
    res = {}
    for mode in ('fused', 'layerwise_bwd'):
        torch.manual_seed(7)
        layer = pcf_layers.PCFLayer(64, 64, cfg(**CHAIN_MODES[mode]), weightnet=[12, 16], num_heads=8,
                                    guidance_feat_len=32).to(device).train()
        with torch.no_grad():
            for m in layer.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.uniform_(-0.3, 0.3)
        x = feats.clone().requires_grad_(True)
        out, _ = layer(xyz, x, idx, nrm)
        out.backward(up)
        res[mode] = dict(out=out.detach(), x=x.grad, **{n: p.grad for n, p in layer.named_parameters()})
    torch.testing.assert_close(res['fused']['out'], res['layerwise_bwd']['out'], rtol=0, atol=0)     # same forward kernels
    top = max(float(t.abs().max()) for k, t in res['layerwise_bwd'].items() if k not in ('out', 'x'))
    for k, v in res['fused'].items():
        ref = res['layerwise_bwd'][k]
        scale = float(ref.abs().max()) + 1e-12
        torch.testing.assert_close(v, ref, rtol=1e-3, atol=1e-3 * scale + 2e-5 * top, msg=lambda m, n=k: f'{n}: {m}')


def test_fused_edge_chain_inference(device):
    """eval(): the chain runs its single inference pass on the running statistics.  Against the layer-at-a-time
    kernels, after one training step so the running statistics moved (they are compared too).  (PCFLayer needs the
    12-channel VI input in the reference as well: mlp_conv is Linear_BN(12, .), layers.py:241.)"""
    import pcf_layers
    torch.manual_seed(3)
    use_vi = True
    B, N, K = 2, 777, 16
    xyz = torch.rand(B, N, 3, device=device)
    nrm = torch.nn.functional.normalize(torch.randn(B, N, 3, device=device), dim=-1)
    nei = torch.cdist(xyz, xyz).topk(K, dim=-1, largest=False).indices.contiguous()
    feats = torch.randn(B, N, 32, device=device)
    outs = {}
    for mode in ('fused', 'off'):
        torch.manual_seed(21)
        wn0 = 12 if use_vi else 3
        layer = pcf_layers.PCFLayer(32, 64, cfg(USE_VI=use_vi, **CHAIN_MODES[mode]), weightnet=[wn0, 16], num_heads=8,
                                    guidance_feat_len=32).to(device)
        if use_vi:
            assert (layer._chain_layers(torch.empty(B, N, K, wn0), nei) is not None) == (mode == 'fused')
        layer.train()
        layer(xyz, feats, nei, nrm)[0].sum().backward()
        layer.eval()
        with torch.no_grad():
            outs[mode] = layer(xyz, feats, nei, nrm)[0]
        rm = {n: b.clone() for n, b in layer.named_buffers() if 'running' in n}
        outs[mode + '_rm'] = rm
    torch.testing.assert_close(outs['fused'], outs['off'], rtol=1e-3, atol=1e-3)
    for n, v in outs['fused_rm'].items():
        torch.testing.assert_close(v, outs['off_rm'][n], rtol=1e-4, atol=1e-5, msg=lambda m, k=n: f'{k}: {m}')


def test_edge_chain_rejects_what_it_does_not_cover(device):
    """Shapes outside the fused kernels: the support predicate says no (callers fall back), and the C entry points
    return PCF_E_UNSUPPORTED with a message instead of launching."""
    import pcf_fused
    assert not pcf_fused.pcf_chain_supported(12, 32, 8, 16, 32, True, 32 * 100)        # K > 16
    assert not pcf_fused.pcf_chain_supported(12, 32, 8, 16, 12, True, 12 * 64)         # K not a power of two
    assert not pcf_fused.pcf_chain_supported(12, 48, 8, 16, 16, True, 16 * 64)         # guidance width > 32
    assert not pcf_fused.pcf_chain_supported(12, 32, 8, 16, 2, True, 2 * 4, edges_per_batch=8)
    lin = [torch.nn.Linear(a, b).to(device) for a, b in [(12, 32), (64, 8), (8, 8), (12, 8), (8, 8), (8, 16)]]
    bns = [torch.nn.BatchNorm1d(l.out_features).to(device) for l in lin]
    vi = torch.randn(1, 8, 32, 12, device=device)
    idx = torch.zeros(1, 8, 32, dtype=torch.long, device=device)
    u = torch.randn(1, 8, 8, device=device)
    fx = torch.randn(1, 8, 16, device=device)
    with pytest.raises(RuntimeError, match='power of two'):
        pcf_fused.pcf_chain(vi, idx, u, fx, list(zip(lin, bns)), True)


def test_eval_mode_backward_against_oracle(device):
    """model.eval() followed by backward (frozen-BatchNorm fine-tuning, saliency): the fused chains need batch statistics in
    their backward, so eval-mode forward under autograd must fall back to the per-layer kernels (running statistics) and
    give the oracle's output, feature gradient and parameter gradients.  Under no_grad the fused inference pass stays on."""
    import pcf_layers
    from oracle import pcf_oracle as O
    g = torch.Generator().manual_seed(12)
    N, K = 384, 16
    xyz = torch.rand(1, N, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=g), dim=-1)
    feats = torch.randn(1, N, 64, generator=g)
    idx = torch.from_numpy(O.knn_bruteforce(xyz[0].numpy(), xyz[0].numpy(), K))[None]
    torch.manual_seed(5)
    layer = pcf_layers.PCFLayer(64, 64, cfg(), weightnet=[12, 16], num_heads=8, guidance_feat_len=32)
    for m in layer.modules():                        # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    layer.to(device).eval()
    d = lambda t: t.to(device)
    assert layer._chain_layers(torch.empty(1, N, K, 12, device=device), d(idx)) is None          # autograd on: no chain
    with torch.no_grad():
        assert layer._chain_layers(torch.empty(1, N, K, 12, device=device), d(idx)) is not None   # inference: fused pass
        out_ng, _ = layer(d(xyz), d(feats), d(idx), d(nrm))
    fd = d(feats).requires_grad_(True)
    out, _ = layer(d(xyz), fd, d(idx), d(nrm))
    up = torch.randn(out.shape, generator=g)
    out.backward(d(up))
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and 'running' not in k)
          for k, v in layer.state_dict().items()}
    fr = feats.clone().requires_grad_(True)
    want, _ = O.pcf_layer(O.Params(sd, '', False), xyz, fr, idx, nrm, num_heads=8)
    want.backward(up)
    torch.testing.assert_close(out.detach().cpu(), want.detach(), **TOL)
    torch.testing.assert_close(out_ng.cpu(), want.detach(), **TOL)
    torch.testing.assert_close(fd.grad.cpu(), fr.grad, **TOL)
    top = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for name, prm in layer.named_parameters():
        torch.testing.assert_close(prm.grad.cpu(), sd[name].grad, rtol=1e-3, atol=1e-3 * max(1.0, top), msg=lambda m, n=name: f'{n}: {m}')
    # a WeightNet alone (PointConv family): same rule
    wn = pcf_layers.WeightNet(12, 16).to(device).eval()
    x = torch.randn(1, 64, 16, 12, device=device)
    y = wn(x)
    y.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in wn.parameters())


@pytest.mark.parametrize('K,N', [(16, 3000), (8, 2048), (4, 1000), (2, 512), (1, 256)])
def test_fused_chain_backward_register_layouts_against_lds_transposes(device, K, N):
    """Pass 3 of the fused edge-graph backward with the outer-product operands produced in registers (transposed lane
    layout, matrix-core operand roles swapped) against the default form that turns them through LDS tiles
    (pcf_cuda.set_chain_backward_engine): feature gradient, du path and all 24 parameter gradients of a PCFLayer, for
    every neighbourhood size the key subtraction distinguishes."""
    import pcf_cuda
    import pcf_layers
    g = torch.Generator().manual_seed(100 + K)
    xyz = torch.rand(2, N, 3, generator=g).to(device)
    nrm = torch.nn.functional.normalize(torch.randn(2, N, 3, generator=g), dim=-1).to(device)
    feats = torch.randn(2, N, 64, generator=g).to(device)
    idx = torch.randint(0, N, (2, N, K), generator=g)
    if K > 1:
        idx[:, :, 0] = torch.arange(N)                  # self first (K = 1 with the self edge alone: an all-zero VI descriptor)
    idx = idx.to(device)
    torch.manual_seed(3)
    layer = pcf_layers.PCFLayer(64, 64, cfg(), weightnet=[12, 16], num_heads=8, guidance_feat_len=32).to(device).train()
    assert layer._chain_layers(torch.empty(2, N, K, 12, device=device), idx) is not None
    up = torch.randn(2, N, 64, generator=g).to(device)
    res = {}
    try:
        for lds in (False, True):
            pcf_cuda.set_chain_backward_engine(lds)
            layer.zero_grad(set_to_none=True)
            f = feats.clone().requires_grad_(True)
            pcf_cuda.launch_log(True)
            out, _ = layer(xyz, f, idx, nrm)
            out.backward(up)
            log = pcf_cuda.read_launch_log()
            assert any(('LDS transposes' if lds else 'register layouts') in k for k in log), log
            res[lds] = [f.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    finally:
        pcf_cuda.launch_log(False)
        pcf_cuda.set_chain_backward_engine(True)
    names = ['feats'] + [n for n, _ in layer.named_parameters()]
    for n, a, b in zip(names, res[False], res[True]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4 * max(1.0, float(b.abs().max())), msg=lambda m, n=n: f'{n}: {m}')
