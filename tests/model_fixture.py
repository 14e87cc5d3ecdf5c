"""Shared by the CPU (oracle) and GPU (HIP) whole-model parity tests: the model_{lite,10cm,2cm} fixtures that
tests/golden/make_golden_models.py produced from the reference model at the real widths of the BASELINE YAMLs."""
import torch

from conftest import (BLOCK_FULL_MAX, BLOCK_SAMPLE, GRAD_FULL_MAX, GRAD_SAMPLE, block_inputs, digest_mismatch, load_golden, split,
                      synthetic_parameters)

TAGS = ('lite', '10cm', '2cm')
LEVELS = 5


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def model_cfg(g, **over):
    """The model keys of the YAML the fixture was made for (from its meta.* entries)."""
    c = Cfg(BATCH_NORM=True, USE_XYZ=True, USE_PE=True, USE_VI=True, point_dim=3, num_level=LEVELS, base_dim=64,
            feat_dim=[int(v) for v in g['meta.feat_dim']], mid_dim=[int(v) for v in g['meta.mid_dim']],
            mid_dim_back=int(g['meta.mid_dim_back']), guided_level=int(g['meta.guided_level']),
            num_heads=int(g['meta.num_heads']), resblocks=[int(v) for v in g['meta.resblocks']],
            resblocks_back=[int(v) for v in g['meta.resblocks_back']], use_level_1=bool(g['meta.use_level_1']),
            drop_path_rate=float(g['meta.drop_path_rate']), num_classes=20, dropout_rate=0., dropout_fc=0.,
            layer_norm_guidance=False, attention_type='subtraction', transformer_type='PCF')
    c.update(over)
    return c


def inputs(g, device='cpu'):
    pcs = [g[f'in.xyz{l}'][None].to(device) for l in range(LEVELS)]
    nrms = [g[f'in.nrm{l}'][None].to(device) for l in range(LEVELS)]
    es = [g[f'in.edges_self{l}'].to(device) for l in range(LEVELS)]
    ef = [g[f'in.edges_forward{l}'].to(device) for l in range(LEVELS - 1)]
    ep = [g[f'in.edges_propagate{l}'].to(device) for l in range(LEVELS - 1)]
    feats = g['in.features'].to(device).clone().requires_grad_(True)
    return feats, pcs, es, ef, ep, nrms


def reference_parameter_names(g):
    return [k[4:] for k in g if k.startswith('gsd.')] + [k[5:] for k in g if k.startswith('gsmp.')]


def drop_factors(g):
    return {k: float(v) for k, v in split(g, 'drop.').items()}


def bad_parameter_grads(named_grads, g, kmap=None, rtol=1e-3, atol=1e-3, noise_mult=4.0):
    """Whole-model parameter gradients against the reference's float64 run, tensor by tensor, with the bound
    rtol / atol * max(1, |ref|_max) widened by noise_mult x the reference's own fp32-vs-fp64 deviation of that tensor
    (`nz.*`).  -> list of offenders.  (The CPU test uses it with noise_mult = 0 for the float64 oracle.)"""
    kmap = kmap or {}
    seen, bad = 0, []
    for name, grad in named_grads:
        ref_name = kmap.get(name, name)
        assert grad is not None, f'{name}: no gradient'
        ref = g['gsd.' + ref_name] if 'gsd.' + ref_name in g else g['gsmp.' + ref_name]
        scale = max(1.0, float(ref.abs().max()))
        floor = noise_mult * float(g['nz.' + ref_name]) / scale
        bad += digest_mismatch(grad, g, '', ref_name, rtol + floor, atol + floor)
        seen += 1
    assert seen == len(reference_parameter_names(g)), (seen, len(reference_parameter_names(g)))
    return bad


def gradient_noise_ratios(named_grads, g, kmap=None, tol=1e-3):
    """Whole-model parameter gradients of an fp32 implementation, measured in units of the reference's OWN fp32 rounding
    deviation: per tensor max|got - truth| / max(nz, tol * scale), where truth = the reference in float64, nz = the
    deviation of the reference's float32 run from it, scale = max(1, |truth|_max).  A faithful fp32 implementation has
    ratios distributed like the reference's own (1 by construction): ~1 in the median, a tail of a few."""
    kmap = kmap or {}
    ratios = {}
    for name, grad in named_grads:
        ref_name = kmap.get(name, name)
        assert grad is not None and torch.isfinite(grad).all(), name
        flat = grad.detach().reshape(-1).float().cpu()
        if 'gsd.' + ref_name in g:
            ref, got = g['gsd.' + ref_name].reshape(-1), flat
        else:
            ref = g['gsmp.' + ref_name]
            got = flat[::flat.numel() // GRAD_SAMPLE][:GRAD_SAMPLE]
        scale = max(1.0, float(ref.abs().max()))
        ratios[name] = float((got - ref).abs().max()) / max(float(g['nz.' + ref_name]), tol * scale)
    assert len(ratios) == len(reference_parameter_names(g))
    return ratios


def block_plan(cfg):
    """[(block name, kind, level in, level out, C in, C out)] in forward order -- the same walk as the generator's
    `block_calls` (model_architecture.py:113-165, :376-398)."""
    plan = []
    fd, base = cfg['feat_dim'], cfg['base_dim']
    if cfg['use_level_1']:
        plan.append(('pcf_backbone.selfpointconv', 'pointconv', 0, 0, 6, base))
        plan.append(('pcf_backbone.selfpointconv_res1', 'self', 0, 0, base, base))
        plan.append(('pcf_backbone.selfpointconv_res2', 'self', 0, 0, base, base))
    for i in range(1, cfg['num_level']):
        plan.append((f'pcf_backbone.pointconv.{i - 1}', 'down', i - 1, i, fd[i - 1], fd[i]))
        for j in range(cfg['resblocks'][i]):
            plan.append((f'pcf_backbone.pointconv_res.{i - 1}.{j}', 'self', i, i, fd[i], fd[i]))
    for i, lvl in enumerate(range(cfg['num_level'] - 2, -1, -1)):
        plan.append((f'pointdeconv.{i}', 'up', lvl + 1, lvl, fd[lvl + 1], base if lvl == 0 else fd[lvl]))
    return plan


def block_case(g, name, kind, lin, lout, cin, cout, device='cpu', dtype=torch.float32):
    """Inputs of one block of fixture g: (feats, skip or None, upstream gradient, edges) on `device`."""
    counts = [g[f'in.xyz{l}'].shape[0] for l in range(LEVELS)]
    feats, skip, up = block_inputs(name, int(g['blk.draw.' + name]), counts[lin], cin, counts[lout], cout, kind == 'up')
    edges = {'pointconv': g[f'in.edges_self{lin}'], 'self': g[f'in.edges_self{lin}'],
             'down': g.get(f'in.edges_forward{lin}'), 'up': g.get(f'in.edges_propagate{lout}')}[kind]
    feats = feats.to(device=device, dtype=dtype).requires_grad_(True)
    skip = skip.to(device=device, dtype=dtype).requires_grad_(True) if skip is not None else None
    return feats, skip, up.to(device=device, dtype=dtype), edges.to(device)


def block_mismatch(g, name, out, feats, skip, named_grads, kmap=None, rtol=1e-3, atol=1e-3):
    """One block's output, input gradient(s) and parameter gradients against its 'blk.*' digests -> offenders."""
    kmap = kmap or {}
    d = dict(full_max=BLOCK_FULL_MAX, sample=BLOCK_SAMPLE)
    bad = digest_mismatch(out, g, 'blk.', 'out.' + name, rtol, atol, **d)
    bad += digest_mismatch(feats.grad, g, 'blk.', 'gin.' + name, rtol, atol, **d)
    if skip is not None:
        bad += digest_mismatch(skip.grad, g, 'blk.', 'gskip.' + name, rtol, atol, **d)
    for k, grad in named_grads:
        ref = name + '.' + kmap.get(k, k)
        assert grad is not None, ref
        # a bias in front of a batch-statistics BatchNorm: analytically zero gradient, rounding noise on every path
        zero_bias = ref.endswith('c.bias') and ('blk.gsd.' + ref[:-6] + 'bn.weight') in g
        # geometry-only sub-networks (inputs: coordinates / VI descriptor, fixed by the model's cloud): their ReLU inputs
        # cannot be kept away from 0 by redrawing the features, and one flipped mask moves one term in N*K (4096 at the
        # 256-point level) of these layers' parameter gradients -- observed up to 1.6e-3 of the scale
        geom = any(t in ref for t in ('.mlp_conv.', '.weightnet.', '.pe_convs.'))
        bad += digest_mismatch(grad, g, 'blk.', ref, rtol, (5.0 if zero_bias else 3.0 if geom else 1.0) * atol, **d)
    return bad


def load(tag):
    return load_golden('model_' + tag)


def synthetic_state(shapes):
    return synthetic_parameters(shapes, seed=7)
