"""Generates tests/golden/{pcf_qk_self,pcf_qk_strided,pcf_ln_self,pcf_ln_strided,ptl_self,ptl_strided}.npz: the
reference's ablation paths
 -- PCFLayer with QK guidance (cfg.attention_type != 'subtraction' -> MultiHeadGuidanceQK,
layers.py:77-114, 264-269) and PointTransformerLayer (layers.py:419-539) -- run through the reference's pure-PyTorch
code exactly as tests/golden/make_golden.py does for the main path (same stand-ins, same helpers).  Build container
only:

    python tests/golden/make_golden_ablation.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG      # noqa: E402  (helpers: shims, clouds, kNN, saving, the layer runner)


def _run_ptl(name, layer, args, gen):
    layer.train()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    args['feats'].requires_grad_(True)
    out = layer(**args)
    gup = torch.randn(out.shape, generator=gen)
    out.backward(gup)
    blobs = {'out.new_feat': out, 'gup': gup}
    for k, v in args.items():
        if v is not None:
            blobs['in.' + k] = v
            if v.grad is not None:
                blobs['gin.' + k] = v.grad
    for k, v in sd.items():
        blobs['sd.' + k] = v
    for k, p in layer.named_parameters():
        if p.grad is not None:
            blobs['gsd.' + k] = p.grad
    MG._save(name, blobs)


def main():
    EasyDict = MG._install_shims()
    import layers
    import model_architecture

    def cfg(**kw):
        c = model_architecture.get_default_configs(EasyDict(), 5, 64)
        c.PCONV_OPT = False
        c.USE_CUDA_KERNEL = False
        for k, v in kw.items():
            c[k] = v
        return c

    gen = torch.Generator().manual_seed(2)
    torch.manual_seed(2)

    N, K = 128, 16
    xyz, nrm = MG._cloud(N, gen)
    idx = MG._knn(xyz, xyz, K)
    lay = layers.PCFLayer(64, 64, cfg(attention_type='qk'), weightnet=[12, 16], num_heads=8, guidance_feat_len=32)
    MG._run_layer('pcf_qk_self', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 64, generator=gen), nei_inds=idx[None], dense_xyz_norm=nrm[None]),
        {'score': (lay.guidance_weight, 'out'), 'w': (lay.weightnet, 'out'), 'agg': (lay.linear, 'in')}, gen,
        dict(num_heads=8, use_vi=1))

    N, M, K = 200, 80, 8
    xyz, nrm = MG._cloud(N, gen)
    sel = torch.randperm(N, generator=gen)[:M]
    sxyz, snrm = xyz[sel] + 0.01 * torch.randn(M, 3, generator=gen), nrm[sel]
    idx = MG._knn(xyz, sxyz, K)
    lay = layers.PCFLayer(32, 64, cfg(attention_type='qk'), weightnet=[12, 4], num_heads=4, guidance_feat_len=32)
    MG._run_layer('pcf_qk_strided', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 32, generator=gen), nei_inds=idx[None], dense_xyz_norm=nrm[None],
        sparse_xyz=sxyz[None], sparse_xyz_norm=snrm[None]),
        {'score': (lay.guidance_weight, 'out'), 'w': (lay.weightnet, 'out'), 'agg': (lay.linear, 'in')}, gen,
        dict(num_heads=4, use_vi=1))

    # ---- cfg.layer_norm_guidance: LayerNorm on query and key in front of the subtraction (layers.py:33-36, 52-53) ----
    N, K = 128, 16
    xyz, nrm = MG._cloud(N, gen)
    idx = MG._knn(xyz, xyz, K)
    lay = layers.PCFLayer(64, 64, cfg(layer_norm_guidance=True), weightnet=[12, 16], num_heads=8, guidance_feat_len=32)
    with torch.no_grad():          # non-trivial affine parameters
        for ln in (lay.guidance_weight.layer_norm_q, lay.guidance_weight.layer_norm_k):
            ln.weight.copy_(torch.rand(ln.weight.shape, generator=gen) + 0.5)
            ln.bias.copy_(torch.randn(ln.bias.shape, generator=gen) * 0.2)
    # 'lin' = the first guidance layer's pre-activation: the draw is repeated until no element sits on the ReLU kink
    MG._run_layer('pcf_ln_self', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 64, generator=gen), nei_inds=idx[None], dense_xyz_norm=nrm[None]),
        {'score': (lay.guidance_weight, 'out'), 'w': (lay.weightnet, 'out'), 'agg': (lay.linear, 'in'),
         'lin': (lay.guidance_weight.mlp[0], 'out')}, gen,
        dict(num_heads=8, use_vi=1))

    N, M, K = 200, 80, 8
    xyz, nrm = MG._cloud(N, gen)
    sel = torch.randperm(N, generator=gen)[:M]
    sxyz, snrm = xyz[sel] + 0.01 * torch.randn(M, 3, generator=gen), nrm[sel]
    idx = MG._knn(xyz, sxyz, K)
    lay = layers.PCFLayer(32, 64, cfg(layer_norm_guidance=True), weightnet=[12, 4], num_heads=4, guidance_feat_len=32)
    MG._run_layer('pcf_ln_strided', lay, dict(
        dense_xyz=xyz[None], dense_feats=torch.randn(1, N, 32, generator=gen), nei_inds=idx[None], dense_xyz_norm=nrm[None],
        sparse_xyz=sxyz[None], sparse_xyz_norm=snrm[None]),
        {'score': (lay.guidance_weight, 'out'), 'w': (lay.weightnet, 'out'), 'agg': (lay.linear, 'in'),
         'lin': (lay.guidance_weight.mlp[0], 'out')}, gen,
        dict(num_heads=4, use_vi=1))

    N, K = 150, 16
    xyz, _ = MG._cloud(N, gen)
    idx = MG._knn(xyz, xyz, K)
    lay = layers.PointTransformerLayer(32, 32, 8)
    _run_ptl('ptl_self', lay, dict(xyz=xyz[None], feats=torch.randn(1, N, 32, generator=gen), nei_ind=idx[None], sparse_xyz=None), gen)

    N, M, K = 240, 90, 12
    xyz, _ = MG._cloud(N, gen)
    sel = torch.randperm(N, generator=gen)[:M]
    sxyz = xyz[sel] + 0.01 * torch.randn(M, 3, generator=gen)
    idx = MG._knn(xyz, sxyz, K)
    lay = layers.PointTransformerLayer(32, 64, 8)
    _run_ptl('ptl_strided', lay, dict(xyz=xyz[None], feats=torch.randn(1, N, 32, generator=gen), nei_ind=idx[None],
                                      sparse_xyz=sxyz[None]), gen)


if __name__ == '__main__':
    main()
