"""Whole-model golden vectors at the REAL widths of the four model YAMLs in BASELINE.json, produced by running the
REFERENCE's pure-PyTorch model (model_architecture.PointConvFormer_Segmentation, USE_CUDA_KERNEL = PCONV_OPT = False)
in the build container.  Companion of make_golden.py (same stand-ins for the absent timm / easydict / pcf_cuda).

    python tests/golden/make_golden_models.py          # rewrites tests/golden/model_{lite,10cm,2cm}.npz

  model_lite   configs/configPCF_10cm_lite.yaml    feat_dim [64,128,192,256,384], mid_dim 4,  resblocks [0,3,3,3,3]
  model_10cm   configs/configPCF_10cm.yaml = configs/configPCF_5cm.yaml (the same network; the YAMLs differ in
               grid_size / batch only)             mid_dim 16, resblocks [0,2,4,6,6], mid_dim_back 1
  model_2cm    configs/configPCF_2cm_PTF2.yaml     use_level_1 False, mid_dim_back 3, resblocks [0,2,4,6,6,2],
               drop_path_rate 0.2 with a FIXED per-block keep mask (timm is absent from the image: the stand-in
               DropPath multiplies by a recorded factor 0 or 1/keep, which is what timm's module does per sample;
               where the factor is applied -- layers.py:414, :739 -- is the reference's own code)

All with K = 16, num_heads = 8 and small clouds (608 / 256 / 112 / 48 / 32 points per level) so the reference runs on
the CPU in seconds.  Parameters are not stored: they are a deterministic function of (name, shape, seed)
(`conftest.synthetic_parameters`), applied to the reference model here and to the build's model in the tests.

Two kinds of content per fixture:

  (A) whole model, the reference run in float64 ("truth"): logits, feature gradient, parameter gradients (whole up to
      8192 elements, else a strided sample + l2 norm and sum) for a random upstream gradient.  The same run in float32
      gives the reference's OWN rounding deviation per tensor (`nz.*`): through 29 BatchNorm-coupled layers the fp32
      gradient of the reference differs from its fp64 gradient by 1-10 % (ReLU masks flipped by 1e-6 perturbations
      are amplified by the backward chain), so no fp32 implementation can meet 1e-3 on whole-model gradients; the
      logits agree to 1e-5.  Tests hold logits to 1e-3 and gradients to the larger of 1e-3 and a multiple of `nz`.
  (B) every block of the model on its own (the model's parameters and neighbourhoods of its level, synthetic feature
      input and upstream gradient drawn from a recorded seed), float64: output sample, input gradient, parameter
      gradients.  A draw is rejected while any ReLU / LeakyReLU input of the block lies within 5e-6 of 0 (gradient
      parity at a kink is a coin toss for any reordered sum).  Tests hold each block to 1e-3.
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import make_golden as MG            # noqa: E402
from conftest import BLOCK_FULL_MAX, BLOCK_SAMPLE, synthetic_parameters, grad_digest, block_inputs      # noqa: E402

YAML = {
    'lite': dict(mid_dim=[4] * 5, mid_dim_back=1, resblocks=[0, 3, 3, 3, 3], use_level_1=True, drop_path_rate=0.),
    '10cm': dict(mid_dim=[16] * 5, mid_dim_back=1, resblocks=[0, 2, 4, 6, 6], use_level_1=True, drop_path_rate=0.),
    '2cm': dict(mid_dim=[16] * 5, mid_dim_back=3, resblocks=[0, 2, 4, 6, 6, 2], use_level_1=False, drop_path_rate=0.2),
}
COUNTS = [608, 256, 112, 48, 32]        # multiples of 16 edges either way (K = 16)
K = 16
KINK_MARGIN = 5e-6


class KinkProbe:
    """Records, per ReLU / LeakyReLU call, the smallest |input| and a checksum of the input (both the reference's
    functional calls and its modules go through torch.nn.functional)."""

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.relu, self.leaky, self.calls = F, F.relu, F.leaky_relu, []

        def note(x):
            d = x.detach()
            self.calls.append((float(d.abs().min()), float(d.double().sum())))

        def relu(x, inplace=False):
            note(x)
            return self.relu(x, inplace=inplace)

        def leaky(x, negative_slope=0.01, inplace=False):
            note(x)
            return self.leaky(x, negative_slope, inplace)

        F.relu, F.leaky_relu = relu, leaky
        return self

    def __exit__(self, *exc):
        self.F.relu, self.F.leaky_relu = self.relu, self.leaky
        return False

    def margin(self, other):
        """Smallest |input| over the calls whose input differs from the same call of `other` (a run on another draw of
        the features): activations that depend on the geometry alone (WeightNet, positional encodings) cannot be moved
        by redrawing, and a flipped mask there changes no feature gradient and one term in ~10^5 of a parameter
        gradient."""
        assert len(self.calls) == len(other.calls)
        m = [a[0] for a, b in zip(self.calls, other.calls) if a[1] != b[1]]
        return min(m) if m else float('inf')


def block_calls(net, c, pcs, nrms, e_self, e_fwd, e_prop):
    """[(block name, module, kind, in_level, out_level, edges)] for every block of the model, in forward order.
    kind: 'pointconv' | 'self' (same-resolution block) | 'down' (strided) | 'up' (transposed)."""
    calls = []
    B = net.pcf_backbone
    if c.use_level_1:
        calls.append(('pcf_backbone.selfpointconv', B.selfpointconv, 'pointconv', 0, 0, e_self[0]))
        calls.append(('pcf_backbone.selfpointconv_res1', B.selfpointconv_res1, 'self', 0, 0, e_self[0]))
        calls.append(('pcf_backbone.selfpointconv_res2', B.selfpointconv_res2, 'self', 0, 0, e_self[0]))
    for i, m in enumerate(B.pointconv):
        calls.append((f'pcf_backbone.pointconv.{i}', m, 'down', i, i + 1, e_fwd[i]))
        for j, r in enumerate(B.pointconv_res[i]):
            calls.append((f'pcf_backbone.pointconv_res.{i}.{j}', r, 'self', i + 1, i + 1, e_self[i + 1]))
    for i, m in enumerate(net.pointdeconv):
        lvl = c.num_level - 2 - i
        calls.append((f'pointdeconv.{i}', m, 'up', lvl + 1, lvl, e_prop[lvl]))
    return calls


def run_block(m, kind, feats, skip, pcs, nrms, lin, lout, edges):
    if kind == 'pointconv':
        return m(pcs[lin][None], feats, edges, nrms[lin][None])[0]
    if kind == 'self':
        return m(pcs[lin][None], feats, edges, nrms[lin][None])[0]
    if kind == 'down':
        return m(pcs[lin][None], feats, edges, nrms[lin][None], pcs[lout][None], nrms[lout][None])[0]
    return m(pcs[lin][None], feats, edges, nrms[lin][None], pcs[lout][None], nrms[lout][None], skip)[0]


def main():
    EasyDict = MG._install_shims()
    import timm.models.layers as tl

    class DropPath(torch.nn.Module):
        """Stand-in for timm's DropPath with the per-sample factor fixed from outside (one sample: packed batch)."""

        def __init__(self, p=0.0):
            super().__init__()
            self.drop_prob, self.factor = p, 1.0

        def forward(self, x):
            return x * self.factor if self.training else x

    tl.DropPath = DropPath
    import model_architecture

    for tag, y in YAML.items():
        gen = torch.Generator().manual_seed({'lite': 21, '10cm': 22, '2cm': 23}[tag])
        c = EasyDict(BATCH_NORM=True, USE_XYZ=True, USE_PE=True, point_dim=3, base_dim=64, feat_dim=[64, 128, 192, 256, 384],
                     guided_level=0, num_heads=8, resblocks_back=[0, 0, 0, 0, 0], num_classes=20, dropout_rate=0., dropout_fc=0.,
                     layer_norm_guidance=False, **y)
        model_architecture.get_default_configs(c, num_level=5, base_dim=64)
        c.PCONV_OPT = False
        c.USE_CUDA_KERNEL = False
        net = model_architecture.PointConvFormer_Segmentation(c)
        net.train()
        params = dict(net.named_parameters())
        shapes = {k: tuple(v.shape) for k, v in params.items()}
        with torch.no_grad():
            for k, v in synthetic_parameters(shapes, seed=7).items():
                params[k].copy_(v)
        drops = {}
        if c.drop_path_rate > 0:
            keep = 1.0 - c.drop_path_rate
            blocks = [n for n, m in net.named_modules() if isinstance(getattr(m, 'drop_path', None), DropPath)]
            for i, n in enumerate(blocks):
                dropped = i in (1, 4, 9, 13) or bool(torch.rand((), generator=gen) < c.drop_path_rate)
                dict(net.named_modules())[n].drop_path.factor = 0.0 if dropped else 1.0 / keep
                drops[n] = 0.0 if dropped else 1.0 / keep
            print(tag, 'drop path:', sum(v == 0 for v in drops.values()), 'of', len(drops), 'blocks dropped')
        xyz0, nrm0 = MG._cloud(COUNTS[0], gen)
        pcs, nrms = [xyz0], [nrm0]
        for n in COUNTS[1:]:
            sel = torch.randperm(pcs[-1].shape[0], generator=gen)[:n]
            pcs.append(pcs[-1][sel])
            nrms.append(nrms[-1][sel])
        e_self = [MG._knn(p, p, K)[None] for p in pcs]
        e_fwd = [MG._knn(pcs[l], pcs[l + 1], K)[None] for l in range(4)]
        e_prop = [MG._knn(pcs[l + 1], pcs[l], K)[None] for l in range(4)]
        feats0 = torch.randn(1, COUNTS[0], 3, generator=gen)
        gup = torch.randn(1, COUNTS[0], c.num_classes, generator=gen)
        blobs = {'gup': gup, 'in.features': feats0}
        for l in range(5):
            blobs[f'in.xyz{l}'], blobs[f'in.nrm{l}'], blobs[f'in.edges_self{l}'] = pcs[l], nrms[l], e_self[l]
        for l in range(4):
            blobs[f'in.edges_forward{l}'], blobs[f'in.edges_propagate{l}'] = e_fwd[l], e_prop[l]

        # ---- (A) whole model: float64 truth, float32 deviation of the reference itself ----
        runs = {}
        for dt in (torch.float64, torch.float32):
            net.to(dt)
            for p in net.parameters():
                p.grad = None
            f = feats0.to(dt).clone().requires_grad_(True)
            out = net(f, [p[None].to(dt) for p in pcs], e_self, e_fwd, e_prop, [n[None].to(dt) for n in nrms])
            out.backward(gup.to(dt))
            runs[dt] = (out.detach().double(), f.grad.double(), {k: p.grad.double().clone() for k, p in net.named_parameters()})
        o64, g64, p64 = runs[torch.float64]
        o32, g32, p32 = runs[torch.float32]
        blobs['out'], blobs['gin.features'] = o64.float(), g64.float()
        blobs['nz.out'] = np.float32((o32 - o64).abs().max())
        blobs['nz.gin.features'] = np.float32((g32 - g64).abs().max())
        worst = 0.0
        for k in shapes:
            blobs.update(grad_digest(k, p64[k].float()))
            blobs['nz.' + k] = np.float32((p32[k] - p64[k]).abs().max())
            worst = max(worst, float((p32[k] - p64[k]).abs().max() / max(1.0, float(p64[k].abs().max()))))
        print(f'{tag}: reference fp32 vs fp64: logits {float(blobs["nz.out"]):.2e}, feature gradient '
              f'{float(blobs["nz.gin.features"] / g64.abs().max()):.2e} rel, worst parameter gradient {worst:.2e} rel')

        # ---- (B) every block on its own, float64, kink-free draws ----
        net.to(torch.float64)
        p64s, n64s = [p.double() for p in pcs], [n.double() for n in nrms]
        for name, m, kind, lin, lout, edges in block_calls(net, c, pcs, nrms, e_self, e_fwd, e_prop):
            cin = m.in_channel
            cout = m.out_channel
            state = {k: v.clone() for k, v in m.named_buffers()}

            def forward(draw):
                feats, skip, up = block_inputs(name, draw, COUNTS[lin], cin, COUNTS[lout], cout, kind == 'up')
                feats = feats.double().requires_grad_(True)
                skip = skip.double().requires_grad_(True) if skip is not None else None
                with torch.no_grad():             # undo the running-statistics update of the previous try
                    for k, b in m.named_buffers():
                        b.copy_(state[k])
                with KinkProbe() as probe:
                    out = run_block(m, kind, feats, skip, p64s, n64s, lin, lout, edges)
                return feats, skip, up, out, probe

            other = forward(100000)[4]
            for draw in range(400):
                feats, skip, up, out, probe = forward(draw)
                if probe.margin(other) >= KINK_MARGIN:
                    break
            else:
                raise RuntimeError(f'{name}: no kink-free draw')
            for p in m.parameters():
                p.grad = None
            out.backward(up.double())
            blobs['blk.draw.' + name] = np.int32(draw)
            bd = lambda k, t: {'blk.' + kk: vv for kk, vv in grad_digest(k, t.float(), BLOCK_FULL_MAX, BLOCK_SAMPLE).items()}
            blobs.update(bd('out.' + name, out.detach()))
            blobs.update(bd('gin.' + name, feats.grad))
            if skip is not None:
                blobs.update(bd('gskip.' + name, skip.grad))
            for k, p in m.named_parameters():
                blobs.update(bd(name + '.' + k, p.grad))
        net.zero_grad()
        for n, f in drops.items():
            blobs['drop.' + n] = np.float32(f)
        for k in ('mid_dim', 'resblocks', 'resblocks_back', 'feat_dim'):
            blobs['meta.' + k] = np.asarray(c[k])
        for k in ('mid_dim_back', 'use_level_1', 'drop_path_rate', 'num_heads', 'guided_level'):
            blobs['meta.' + k] = np.asarray(float(c[k]))
        blobs['meta.n_params'] = np.asarray(sum(p.numel() for p in net.parameters()))
        MG._save('model_' + tag, blobs)


if __name__ == '__main__':
    main()
