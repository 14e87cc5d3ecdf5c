"""Generate the golden vectors under tests/golden/ by running the REFERENCE's pure-PyTorch path.

Runs only in the build container (needs /root/reference); the GPU box sees just the .npz
files this writes.  The reference is imported as-is with three in-memory stand-ins for
packages absent from the image: timm's DropPath (identity at rate
0), an empty ``pcf_cuda`` (never called: USE_CUDA_KERNEL=False), and easydict.  Nothing
from the reference is copied: the fixtures are inputs, parameters and the outputs /
autograd gradients the reference computed for them.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Each layer fixture holds: ``in.*`` call arguments, ``sd.*`` the module state_dict BEFORE
the forward, ``out.*`` outputs, ``gup`` the random upstream gradient, ``gin.*`` input
gradients, ``gsd.*`` parameter gradients and ``cap.*`` / ``gcap.*`` tensors captured at the
pcf_cuda operator boundary inside the layer (and their gradients).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def _install_shims():
    class DropPath(torch.nn.Identity):
        def __init__(self, p=0.0):
            super().__init__()

    for name in ('timm', 'timm.models', 'timm.models.layers'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['timm.models.layers'].DropPath = DropPath
    sys.modules['pcf_cuda'] = types.ModuleType('pcf_cuda')

    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    ed = types.ModuleType('easydict')
    ed.EasyDict = EasyDict
    sys.modules['easydict'] = ed
    sys.path.insert(0, REF)
    return EasyDict


def _cloud(n, gen, scale=1.0):
    xyz = torch.rand(n, 3, generator=gen) * scale
    nrm = torch.randn(n, 3, generator=gen)
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    return xyz, nrm


def _knn(ref, query, K):
    from sklearn.neighbors import KDTree
    idx = KDTree(ref.numpy()).query(query.numpy(), k=K, return_distance=False)
    return torch.from_numpy(idx.astype(np.int64))


def _save(name, blobs):
    flat = {}
    for k, v in blobs.items():
        if v is None:
            continue
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        flat[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **flat)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(flat)} arrays')


KINK_MARGIN = 1e-4


def _run_layer(name, layer, args, captures, gen, meta):
    """args: ordered dict of forward kwargs (tensors or None); captures: {tag: (module, 'in'|'out')}.

    The gradient of a ReLU is discontinuous at 0: a fixture whose LAST pre-activation (the captured
    ``lin`` tensor, when there is one) has an element within fp32 noise of 0 would make gradient parity
    a coin toss for any implementation that sums in a different order.  Such a draw is rejected: the
    feature tensor is redrawn until every element of ``lin`` is at least KINK_MARGIN away from 0."""
    if 'lin' in captures:
        mod, which = captures['lin']
        feat_key = next(k for k in args if k.endswith('feats'))
        for attempt in range(50):
            seen = {}
            h = mod.register_forward_hook(lambda m, i, o: seen.__setitem__('lin', o.detach().clone()))   # before any in-place ReLU
            layer.train()
            state = {k: v.clone() for k, v in layer.state_dict().items()}
            with torch.no_grad():
                layer(**args)
            layer.load_state_dict(state)          # undo the running-stat update of the probe
            h.remove()
            if float(seen['lin'].abs().min()) >= KINK_MARGIN:
                break
            args[feat_key] = torch.randn(args[feat_key].shape, generator=gen)
        else:
            raise RuntimeError(f'{name}: no kink-free draw found')
        if attempt:
            print(f'{name}: redrew features {attempt}x to stay {KINK_MARGIN} away from the last ReLU kink')
    layer.train()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    caps = {}

    def grab(tag, which):
        def hook(mod, inp, out):
            t = inp[0] if which == 'in' else out
            if t.requires_grad:
                t.retain_grad()
            caps[tag] = t
        return hook

    for tag, (mod, which) in captures.items():
        mod.register_forward_hook(grab(tag, which))
    for k, v in args.items():
        if isinstance(v, torch.Tensor) and v.is_floating_point() and k.endswith('feats'):
            v.requires_grad_(True)
    out, wn_in = layer(**args)
    gup = torch.randn(out.shape, generator=gen)
    out.backward(gup)
    blobs = {'out.new_feat': out, 'out.wn_in': wn_in, 'gup': gup}
    for k, v in args.items():
        blobs['in.' + k] = v
        if isinstance(v, torch.Tensor) and v.grad is not None:
            blobs['gin.' + k] = v.grad
    for k, v in sd.items():
        blobs['sd.' + k] = v
    for k, p in layer.named_parameters():
        if p.grad is not None:
            blobs['gsd.' + k] = p.grad
    for tag, t in caps.items():
        blobs['cap.' + tag] = t
        if t.grad is not None:
            blobs['gcap.' + tag] = t.grad
    for k, v in meta.items():
        blobs['meta.' + k] = np.asarray(v)
    _save(name, blobs)


def main():
    EasyDict = _install_shims()
    import layers
    import model_architecture

    def cfg(**kw):
        c = model_architecture.get_default_configs(EasyDict(), 5, 64)
        c.PCONV_OPT = False
        c.USE_CUDA_KERNEL = False
        for k, v in kw.items():
            c[k] = v
        return c

    gen = torch.Generator().manual_seed(1)
    torch.manual_seed(1)

    # ---- PCFLayer, self neighbourhood, BASELINE channel shape (64->64, H=8, Cm=16) ----
    N, K = 160, 16
    xyz, nrm = _cloud(N, gen)
    idx = _knn(xyz, xyz, K)
    lay = layers.PCFLayer(64, 64, cfg(), weightnet=[12, 16], num_heads=8, guidance_feat_len=32)
    _run_layer('pcf_self_64', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 64, generator=gen), nei_inds=idx[None],
        dense_xyz_norm=nrm[None]),
        {'fx': (lay.unary1, 'out'), 'score': (lay.guidance_weight, 'out'), 'w': (lay.weightnet, 'out'),
         'agg': (lay.linear, 'in')}, gen, dict(num_heads=8, use_vi=1))

    # ---- PCFLayer, self, channel change (32->64): unary1 + unary_shortcut present ----
    N, K = 128, 16
    xyz, nrm = _cloud(N, gen)
    idx = _knn(xyz, xyz, K)
    lay = layers.PCFLayer(32, 64, cfg(), weightnet=[12, 16], num_heads=8, guidance_feat_len=32)
    _run_layer('pcf_self_32_64', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 32, generator=gen), nei_inds=idx[None],
        dense_xyz_norm=nrm[None]),
        {'fx': (lay.unary1, 'out'), 'score': (lay.guidance_weight, 'out'), 'w': (lay.weightnet, 'out'),
         'agg': (lay.linear, 'in')}, gen, dict(num_heads=8, use_vi=1))

    # ---- PCFLayer, strided (N=300 -> M=100), K=8, H=4, Cm=4 ----
    N, M, K = 300, 100, 8
    xyz, nrm = _cloud(N, gen)
    sel = torch.randperm(N, generator=gen)[:M]
    sxyz, snrm = xyz[sel] + 0.01 * torch.randn(M, 3, generator=gen), nrm[sel]
    idx = _knn(xyz, sxyz, K)
    lay = layers.PCFLayer(32, 64, cfg(), weightnet=[12, 4], num_heads=4, guidance_feat_len=32)
    _run_layer('pcf_strided', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 32, generator=gen), nei_inds=idx[None],
        dense_xyz_norm=nrm[None], sparse_xyz=sxyz[None], sparse_xyz_norm=snrm[None]),
        {'fx': (lay.unary1, 'out'), 'score': (lay.guidance_weight, 'out'), 'w': (lay.weightnet, 'out'),
         'agg': (lay.linear, 'in')}, gen, dict(num_heads=4, use_vi=1))

    # ---- PointConv as in test_configs/pointconv_single.yaml: 3->32, weightnet [3,16], no BN/PE/VI ----
    N, K = 256, 16
    xyz, nrm = _cloud(N, gen)
    idx = _knn(xyz, xyz, K)
    c = cfg(BATCH_NORM=False, USE_PE=False, USE_VI=False)
    lay = layers.PointConv(3, 32, c, weightnet=[3, 16])
    _run_layer('pointconv_single', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 3, generator=gen), nei_inds=idx[None]),
        {'w': (lay.weightnet, 'out'), 'agg': (lay.linear, 'in'), 'lin': (lay.linear, 'out')}, gen,
        dict(use_vi=0, use_pe=0))

    # ---- PointConv level-0 of configPCF_10cm: 6->64, VI + PE (Ci=6, Ca=12, Cm=16), BN ----
    N, K = 144, 16
    xyz, nrm = _cloud(N, gen)
    idx = _knn(xyz, xyz, K)
    c = cfg(USE_PE=True)
    lay = layers.PointConv(6, 64, c, weightnet=[12, 16])
    _run_layer('pointconv_vi_pe', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 6, generator=gen), nei_inds=idx[None],
        dense_xyz_norm=nrm[None]),
        {'w': (lay.weightnet, 'out'), 'agg': (lay.linear, 'in'), 'lin': (lay.linear.c, 'out')}, gen,
        dict(use_vi=1, use_pe=1))

    # ---- PointConvStridePE 64->64 (Ci=16, Ca=16), strided N=240 -> M=90, K=12 ----
    N, M, K = 240, 90, 12
    xyz, nrm = _cloud(N, gen)
    sel = torch.randperm(N, generator=gen)[:M]
    sxyz, snrm = xyz[sel], nrm[sel]
    idx = _knn(xyz, sxyz, K)
    lay = layers.PointConvStridePE(64, 64, cfg(USE_PE=True), weightnet=[12, 16])
    _run_layer('stride_pe', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 64, generator=gen), nei_inds=idx[None],
        dense_xyz_norm=nrm[None], sparse_xyz=sxyz[None], sparse_xyz_norm=snrm[None]),
        {'fx': (lay.unary1, 'out'), 'pe': (lay.pe_convs, 'out'), 'w': (lay.weightnet, 'out'),
         'agg': (lay.linear, 'in'), 'lin': (lay.linear.c, 'out')}, gen, dict(use_vi=1))

    # ---- PointConvTransposePE 128->64 (Ci=128, Ca=16, Cm=1), sparse N=60 -> dense M=200 ----
    N, M, K = 60, 200, 16
    dxyz, dnrm = _cloud(M, gen)
    sel = torch.randperm(M, generator=gen)[:N]
    sxyz, snrm = dxyz[sel], dnrm[sel]
    idx = _knn(sxyz, dxyz, K)
    lay = layers.PointConvTransposePE(128, 64, cfg(USE_PE=True), weightnet=[12, 1], mlp2=[64, 64])
    _run_layer('transpose_pe', lay, dict(
        sparse_xyz=sxyz[None], sparse_feats=torch.randn(1, N, 128, generator=gen), nei_inds=idx[None],
        sparse_xyz_norm=snrm[None], dense_xyz=dxyz[None], dense_xyz_norm=dnrm[None],
        dense_feats=torch.randn(1, M, 64, generator=gen)),
        {'pe': (lay.pe_convs, 'out'), 'w': (lay.weightnet, 'out'), 'agg': (lay.linear, 'in'),
         'lin': (lay.linear.c, 'out')}, gen, dict(use_vi=1, use_pe=1))

    # ---- whole segmentation model (backbone + decoder), 3 levels, tiny widths ----
    mcfg = cfg(USE_PE=True, num_classes=5, dropout_fc=0.)
    model_architecture.get_default_configs(mcfg, num_level=3, base_dim=16)
    mcfg.feat_dim = [16, 32, 48]
    mcfg.mid_dim = [4, 4, 4]
    mcfg.mid_dim_back = 1
    mcfg.guided_level = 0
    mcfg.num_heads = 4
    mcfg.resblocks = [0, 2, 1]
    mcfg.resblocks_back = [0, 0, 0]
    mcfg.PCONV_OPT = False
    mcfg.USE_CUDA_KERNEL = False
    torch.manual_seed(11)
    net = model_architecture.PointConvFormer_Segmentation(mcfg)
    net.train()
    counts, K = [240, 90, 36], 8
    xyz0, nrm0 = _cloud(counts[0], gen)
    pcs, nrms = [xyz0], [nrm0]
    for c in counts[1:]:
        sel = torch.randperm(pcs[-1].shape[0], generator=gen)[:c]
        pcs.append(pcs[-1][sel])
        nrms.append(nrms[-1][sel])
    e_self = [_knn(p, p, K)[None] for p in pcs]
    e_fwd = [_knn(pcs[l], pcs[l + 1], K)[None] for l in range(len(pcs) - 1)]
    e_prop = [_knn(pcs[l + 1], pcs[l], K)[None] for l in range(len(pcs) - 1)]
    feats = torch.randn(1, counts[0], 3, generator=gen).requires_grad_(True)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    out = net(feats, [p[None] for p in pcs], e_self, e_fwd, e_prop, [n[None] for n in nrms])
    gup = torch.randn(out.shape, generator=gen)
    out.backward(gup)
    blobs = {'out': out, 'gup': gup, 'in.features': feats, 'gin.features': feats.grad}
    for l in range(len(pcs)):
        blobs[f'in.xyz{l}'] = pcs[l]
        blobs[f'in.nrm{l}'] = nrms[l]
        blobs[f'in.edges_self{l}'] = e_self[l]
    for l in range(len(pcs) - 1):
        blobs[f'in.edges_forward{l}'] = e_fwd[l]
        blobs[f'in.edges_propagate{l}'] = e_prop[l]
    for k, v in sd.items():
        blobs['sd.' + k] = v
    for k, p_ in net.named_parameters():
        blobs['gsd.' + k] = p_.grad
    _save('model_seg3', blobs)

    # ---- VI transform alone, including a zero offset (self edge) ----
    import layer_utils
    N, K = 96, 16
    xyz, nrm = _cloud(N, gen)
    idx = _knn(xyz, xyz, K)
    rel = xyz[idx] - xyz[:, None]
    vi = layer_utils.VI_coordinate_transform(rel[None], nrm[idx][None], nrm[None], K)
    _save('vi_transform', {'xyz': xyz, 'nrm': nrm, 'idx': idx, 'vi': vi[0]})

    # ---- kNN: the reference's default CPU engine (sklearn KDTree, datasetCommon.py:115-120) ----
    # Stored with the K+1-th distance gap so the test can assert "equal wherever no tie".
    ref, _ = _cloud(700, gen, scale=3.0)
    qry, _ = _cloud(333, gen, scale=3.0)
    for tag, r, q, K in (('self', ref, ref, 16), ('cross', ref, qry, 16), ('k5', qry, ref, 5)):
        from sklearn.neighbors import KDTree
        d, i = KDTree(r.numpy()).query(q.numpy(), k=K + 1, return_distance=True)
        _save('knn_' + tag, {'ref': r, 'query': q, 'idx': i[:, :K].astype(np.int64),
                             'dist': d.astype(np.float64)})


if __name__ == '__main__':
    main()
