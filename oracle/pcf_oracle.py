"""CPU oracle for the PointConvFormer hot path  --  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-on-CPU (fp32, or fp64 when asked), the maths of the
reference's hot path.  It is the checker for the HIP kernels, never the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``ml-pointconvformer_amd/`` imports it.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py``
against golden vectors produced in the build container by importing the reference's
pure-PyTorch path (``tests/golden/make_golden.py``).  kNN is pinned
by definition (squared L2 in difference form, ascending, ties -> lower index) and
cross-checked against sklearn KDTree when the goldens are generated, because the reference
holds no fixture for it.

Each function cites the reference file:line it follows (paths relative to the reference
root).  The code is written from the formulas, not transcribed.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# --------------------------------------------------------------------------------------
# gathers / geometry
# --------------------------------------------------------------------------------------
def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """table [B,N,C], idx [B,M,K] (or [B,M]) -> [B,M,K,C].   (layer_utils.py:13-30)"""
    B = table.shape[0]
    flat = idx.reshape(B, -1)
    out = torch.stack([table[b].index_select(0, flat[b]) for b in range(B)], 0)
    return out.reshape(*idx.shape, table.shape[-1])


def _unit(v: torch.Tensor) -> torch.Tensor:
    # F.normalize semantics: v / max(||v||, 1e-12)   (layer_utils.py:190,197,199)
    n = v.pow(2).sum(-1, keepdim=True).sqrt().clamp_min(1e-12)
    return v / n


def _cross(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    ax, ay, az = a.unbind(-1)
    bx, by, bz = b.unbind(-1)
    return torch.stack([ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx], -1)


def vi_features(rel: torch.Tensor, nbr_normal: torch.Tensor, ctr_normal: torch.Tensor) -> torch.Tensor:
    """Viewpoint-invariant edge descriptor, 12 channels.   (layer_utils.py:176-231)

    rel [B,M,K,3] = neighbour xyz - centre xyz; nbr_normal [B,M,K,3]; ctr_normal [B,M,3].
    Channels: n_j.n_i, r^.n_i, r^.n_j, r.n_i, r^.n_j (again), n_j.v, n_j.w, r.(n_j x n_i), |r|, r.
    """
    n_i = ctr_normal.unsqueeze(2)
    r_hat = _unit(rel)
    proj = (r_hat * n_i).sum(-1, keepdim=True)
    v = _unit(n_i - proj * r_hat)
    w = _unit(_cross(r_hat, v))
    dot = lambda a, b: (a * b).sum(-1, keepdim=True)
    t1 = dot(nbr_normal, n_i)
    t2 = proj
    t3 = dot(r_hat, nbr_normal)
    t4 = dot(rel, n_i)
    t5 = t3
    t6 = dot(nbr_normal, v)
    t7 = dot(nbr_normal, w)
    t8 = dot(rel, _cross(nbr_normal, n_i.expand_as(nbr_normal)))
    t9 = rel.pow(2).sum(-1, keepdim=True).sqrt()
    return torch.cat([t1, t2, t3, t4, t5, t6, t7, t8, t9, rel], -1)


# --------------------------------------------------------------------------------------
# aggregate operators at the pcf_cuda boundary
# --------------------------------------------------------------------------------------
def _edge_tile(x, idx, guid=None, add=None):
    """[B,M,K,Ci(+Ca)]: gathered rows, optionally head-modulated, with `add` appended."""
    t = gather_rows(x, idx)
    if guid is not None:
        H = guid.shape[-1]
        Ci = x.shape[-1]
        head = torch.arange(Ci) % H                      # head of channel c is c % H (layers.py:387-388)
        t = t * guid[..., head]
    if add is not None and add.shape[-1] > 0:
        t = torch.cat([t, add], -1)
    return t


def pcf_forward(x, idx, guid, w):
    """out[b,n,c*Cm+m] = sum_k x[b,idx[b,n,k],c] * guid[b,n,k,c%H] * w[b,n,k,m].

    (pcf_ops.cu:27-71; layers.py:387-390)"""
    t = _edge_tile(x, idx, guid)
    out = torch.einsum('bnkc,bnkm->bncm', t, w)
    return out.reshape(out.shape[0], out.shape[1], -1)


def pcf_backward(gout, x, idx, guid, w):
    """Adjoint of pcf_forward in the FORWARD channel layout.  Returns
    (grad_x, grad_guid, grad_w) -- what torch.autograd gives for layers.py:387-390."""
    B, M, K = idx.shape
    Ci, Cm, H = x.shape[-1], w.shape[-1], guid.shape[-1]
    g = gout.reshape(B, M, Ci, Cm)
    xg = gather_rows(x, idx)                              # [B,M,K,Ci]
    head = torch.arange(Ci) % H
    gt = torch.einsum('bncm,bnkm->bnkc', g, w)            # d/d(tile)
    grad_w = torch.einsum('bncm,bnkc->bnkm', g, xg * guid[..., head])
    grad_guid = torch.zeros_like(guid)
    grad_guid.index_add_(3, head, gt * xg)
    contrib = gt * guid[..., head]
    grad_x = torch.zeros_like(x)
    for b in range(B):
        grad_x[b].index_add_(0, idx[b].reshape(-1), contrib[b].reshape(-1, Ci))
    return grad_x, grad_guid, grad_w


def pconv_forward(x, idx, w, add):
    """out[b,n,c*Cm+m] = sum_k cat(x[b,idx[b,n,k]], add[b,n,k])[c] * w[b,n,k,m].

    (pconv_ops.cu:40-103; layers.py:890-897)"""
    t = _edge_tile(x, idx, None, add)
    out = torch.einsum('bnkc,bnkm->bncm', t, w)
    return out.reshape(out.shape[0], out.shape[1], -1)


def pconv_backward(gout, x, idx, w, add):
    """(grad_x, grad_w, grad_add): autograd adjoint of pconv_forward (forward layout, F1)."""
    B, M, K = idx.shape
    Ci, Ca, Cm = x.shape[-1], add.shape[-1], w.shape[-1]
    g = gout.reshape(B, M, Ci + Ca, Cm)
    t = _edge_tile(x, idx, None, add)
    grad_w = torch.einsum('bncm,bnkc->bnkm', g, t)
    gt = torch.einsum('bncm,bnkm->bnkc', g, w)
    grad_add = gt[..., Ci:].contiguous()
    grad_x = torch.zeros_like(x)
    for b in range(B):
        grad_x[b].index_add_(0, idx[b].reshape(-1), gt[b, ..., :Ci].reshape(-1, Ci))
    return grad_x, grad_w, grad_add


def pconv_linear_forward(x, idx, w, add, lin_w, lin_b):
    """(out [B,M,Co], pconv_out [B,M,Cm(Ci+Ca)]).   (pconv_ops.cu:129-225, 969-1269;
    layers.py:894-898)"""
    p = pconv_forward(x, idx, w, add)
    return p @ lin_w.t() + lin_b, p


def pconv_linear_backward(gout, x, idx, w, add, lin_w, pconv_out):
    """(grad_x, grad_w, grad_add, grad_lin_w, grad_lin_b): adjoint of pconv_linear_forward."""
    Co = gout.shape[-1]
    g2 = gout.reshape(-1, Co)
    grad_lin_w = g2.t() @ pconv_out.reshape(g2.shape[0], -1)
    grad_lin_b = g2.sum(0)
    gp = gout @ lin_w
    gx, gw, ga = pconv_backward(gp, x, idx, w, add)
    return gx, gw, ga, grad_lin_w, grad_lin_b


# --------------------------------------------------------------------------------------
# kNN and its CSR transpose
# --------------------------------------------------------------------------------------
def knn_bruteforce(ref: np.ndarray, query: np.ndarray, K: int) -> np.ndarray:
    """K nearest refs per query: squared L2 in difference form ((rx-qx)^2+(ry-qy)^2)+(rz-qz)^2
    evaluated in fp32 with one rounding per operation, ascending, ties -> lower ref index.

    Mirrors what knn_post_dataloader_utils.py:22-41 asks of KeOps (argKmin of the same
    expression).  numpy version for small inputs; oracle/knn_ref.c is the same definition
    in C for large ones.  Returns int64 [Nq,K]."""
    ref = np.ascontiguousarray(ref, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    out = np.empty((query.shape[0], K), np.int64)
    step = max(1, (1 << 24) // max(1, ref.shape[0]))
    for s in range(0, query.shape[0], step):
        q = query[s:s + step]
        dx = ref[None, :, 0] - q[:, None, 0]
        dy = ref[None, :, 1] - q[:, None, 1]
        dz = ref[None, :, 2] - q[:, None, 2]
        d = (dx * dx + dy * dy) + dz * dz
        out[s:s + step] = np.argsort(d, axis=1, kind='stable')[:, :K]
    return out


def knn_packed(ref, query, ref_offsets, query_offsets, K):
    """Per-sample kNN over a packed batch; indices are global (offset into the packed ref).
    (knn_post_dataloader_utils.py:171-223 control flow + :113-154 offsetting)"""
    out = np.empty((query.shape[0], K), np.int64)
    for s in range(len(ref_offsets) - 1):
        r0, r1 = ref_offsets[s], ref_offsets[s + 1]
        q0, q1 = query_offsets[s], query_offsets[s + 1]
        out[q0:q1] = knn_bruteforce(ref[r0:r1], query[q0:q1], K) + r0
    return out


def knn_inverse(idx: np.ndarray, total_points: int):
    """CSR transpose of a [Nq,K] neighbour table, buckets in (query, k) order.

    (knn.cu:104-168; test_kernels.py:177-213).  Entries outside [0,total_points) are skipped
    (knn.cu:38,76).  Returns inv_neighbors int32 [Nq*K], inv_k uint8 [Nq*K], inv_idx int32
    [total_points+1]; unused tail of the first two stays 0 as torch::zeros leaves it."""
    Nq, K = idx.shape
    flat = idx.reshape(-1)
    valid = (flat >= 0) & (flat < total_points)
    edge = np.nonzero(valid)[0]
    order = np.argsort(flat[edge], kind='stable')
    edge = edge[order]
    inv_neighbors = np.zeros(Nq * K, np.int32)
    inv_k = np.zeros(Nq * K, np.uint8)
    inv_neighbors[:edge.size] = (edge // K).astype(np.int32)
    inv_k[:edge.size] = (edge % K).astype(np.uint8)
    counts = np.bincount(flat[edge], minlength=total_points)[:total_points]
    inv_idx = np.zeros(total_points + 1, np.int32)
    np.cumsum(counts, out=inv_idx[1:])
    return inv_neighbors, inv_k, inv_idx


# --------------------------------------------------------------------------------------
# layer building blocks (functional; parameters come in a dict with the reference's names)
# --------------------------------------------------------------------------------------
class Params:
    """Prefix view over a flat {name: tensor} dict (a reference state_dict)."""

    def __init__(self, table, prefix='', training=True, update_running=False):
        self.t, self.p, self.training, self.update_running = table, prefix, training, update_running

    def sub(self, name):
        return Params(self.t, f'{self.p}{name}.', self.training, self.update_running)

    def has(self, name):
        return f'{self.p}{name}' in self.t

    def __getitem__(self, name):
        return self.t[f'{self.p}{name}']


def batchnorm_lastdim(y, P: Params, momentum=0.1):
    """BatchNorm over every axis but the last.  Training: batch statistics, biased variance
    (what F.batch_norm applies); eval: running statistics.  (layer_utils.py:272-277,
    util/cp_batchnorm.py:13-30)"""
    C = y.shape[-1]
    flat = y.reshape(-1, C)
    if P.training:
        mean = flat.mean(0)
        var = flat.var(0, unbiased=False)
        if P.update_running and P.has('running_mean'):
            n = flat.shape[0]
            with torch.no_grad():
                P['running_mean'].mul_(1 - momentum).add_(momentum * mean)
                P['running_var'].mul_(1 - momentum).add_(momentum * var * n / max(n - 1, 1))
    else:
        mean, var = P['running_mean'], P['running_var']
    return (y - mean) * torch.rsqrt(var + BN_EPS) * P['weight'] + P['bias']


def linear_bn(x, P: Params):
    """Linear_BN: y = BN(x W^T + b).   (layer_utils.py:241-277)"""
    return batchnorm_lastdim(F.linear(x, P['c.weight'], P['c.bias']), P.sub('bn'))


def maybe_linear_bn(x, P: Params):
    """Linear_BN when the dict holds one, plain nn.Linear otherwise (cfg.BATCH_NORM False)."""
    if P.has('c.weight'):
        return linear_bn(x, P)
    return F.linear(x, P['weight'], P['bias'])


def unary_block(x, P: Params, relu=True):
    """UnaryBlock = Linear_BN(1d) [+ LeakyReLU(0.1)].   (layer_utils.py:281-315)"""
    y = maybe_linear_bn(x, P.sub('mlp'))
    return F.leaky_relu(y, 0.1) if relu else y


def weightnet(x, P: Params):
    """WeightNet: (Linear_BN + ReLU) for every layer, the last included.  (layers.py:163-171)"""
    i = 0
    while P.has(f'mlp_convs.{i}.c.weight'):
        x = F.relu(linear_bn(x, P.sub(f'mlp_convs.{i}')))
        i += 1
    return x


def guidance_scores(query, key, P: Params):
    """MultiHeadGuidance, subtraction form: sigmoid(MLP(q - k)), ReLU between layers.
    (layers.py:47-68; with cfg.layer_norm_guidance a LayerNorm on the query and another on the key first)"""
    if P.has('layer_norm_q.weight'):          # cfg.layer_norm_guidance (layers.py:33-36, 52-53)
        C = query.shape[-1]
        query = F.layer_norm(query, (C,), P['layer_norm_q.weight'], P['layer_norm_q.bias'])
        key = F.layer_norm(key, (C,), P['layer_norm_k.weight'], P['layer_norm_k.bias'])
    s = query - key
    n = 0
    while P.has(f'mlp.{n}.c.weight') or P.has(f'mlp.{n}.weight'):
        n += 1
    for i in range(n):
        s = maybe_linear_bn(s, P.sub(f'mlp.{i}'))
        s = torch.sigmoid(s) if i == n - 1 else F.relu(s)
    return s


def guidance_scores_qk(query, key, P: Params, num_heads):
    """MultiHeadGuidanceQK: sigmoid(scale * <W q, W k>) per head, the key being the neighbourhood's single centre row.
    (layers.py:77-114)  The reference runs the SAME Linear_BN over the query tensor and over the key tensor repeated
    K times (two BatchNorm calls, each with its own batch statistics); only key row 0 of the result is used."""
    B, N, K, _ = query.shape
    q = linear_bn(query, P.sub('qk_linear'))
    k = linear_bn(key.expand(-1, -1, K, -1), P.sub('qk_linear'))
    d = q.shape[-1] // num_heads
    q = q.view(B, N, K, num_heads, d)
    k = k.view(B, N, K, num_heads, d)[:, :, :1]
    return torch.sigmoid((q * k).sum(-1) * d ** -0.5)


def _geometry(dense_xyz, dense_norm, idx, sparse_xyz, sparse_norm, use_vi, vi=None):
    ctr_xyz = dense_xyz if sparse_xyz is None else sparse_xyz
    ctr_norm = dense_norm if sparse_norm is None else sparse_norm
    rel = gather_rows(dense_xyz, idx) - ctr_xyz.unsqueeze(2)
    if not use_vi:
        return rel, rel
    if vi is not None:
        return rel, vi
    return rel, vi_features(rel, gather_rows(dense_norm, idx), ctr_norm)


def pcf_layer(P: Params, dense_xyz, dense_feats, idx, dense_norm, sparse_xyz=None, sparse_norm=None,
              vi=None, num_heads=8, use_vi=True, drop_scale=None):
    """PCFLayer.forward.   (layers.py:306-416)  Returns (new_feat, weightNetInput).  drop_scale: the factor DropPath
    drew for the residual branch (0 or 1/keep; timm DropPath with one sample per packed batch), None = identity."""
    B, N, _ = dense_xyz.shape
    M = N if sparse_xyz is None else sparse_xyz.shape[1]
    K = idx.shape[2]
    fx = unary_block(dense_feats, P.sub('unary1')) if P.has('unary1.mlp.c.weight') else dense_feats
    _, wn_in = _geometry(dense_xyz, dense_norm, idx, sparse_xyz, sparse_norm, use_vi, vi)
    pe = F.relu(maybe_linear_bn(wn_in, P.sub('mlp_conv')))
    gx = unary_block(fx, P.sub('guidance_unary'), relu=False)
    q = torch.cat([gather_rows(gx, idx), pe], -1)
    key = q[:, :, :1] if M == N else q.max(2, keepdim=True)[0]
    if P.has('guidance_weight.qk_linear.c.weight'):         # cfg.attention_type != 'subtraction' (layers.py:264-269)
        score = guidance_scores_qk(q, key, P.sub('guidance_weight'), num_heads)
    else:
        score = guidance_scores(q, key, P.sub('guidance_weight'))
    w = weightnet(wn_in, P.sub('weightnet'))
    agg = pcf_forward(fx, idx, score, w)
    y = F.relu(maybe_linear_bn(agg, P.sub('linear')))
    y = unary_block(y, P.sub('unary2'), relu=False)
    short = dense_feats if sparse_xyz is None else gather_rows(dense_feats, idx).max(2)[0]
    if P.has('unary_shortcut.mlp.c.weight'):
        short = unary_block(short, P.sub('unary_shortcut'), relu=False)
    if drop_scale is not None:          # layers.py:414
        y = y * drop_scale
    return F.leaky_relu(y + short, 0.1), wn_in


def point_transformer_layer(P: Params, xyz, feats, idx, sparse_xyz=None, share_planes=8):
    """PointTransformerLayer.forward (the ablation block, layers.py:419-539): vector attention over the K neighbours
    with a softmax, positional term from the coordinate offsets.  Returns new_feats [1, M, out_planes]."""
    K = idx.shape[2]
    fq = F.linear(feats, P['linear_q.weight'], P['linear_q.bias'])
    fk = gather_rows(F.linear(feats, P['linear_k.weight'], P['linear_k.bias']), idx)[0]       # [M, K, mid]
    fv = gather_rows(F.linear(feats, P['linear_v.weight'], P['linear_v.bias']), idx)[0]       # [M, K, out]
    if sparse_xyz is not None:
        dxyz = gather_rows(xyz, idx) - sparse_xyz[:, :, None]
        fq = gather_rows(fq, idx[:, :, :1])                                                   # [1, M, 1, mid]
    else:
        dxyz = gather_rows(xyz, idx) - xyz[:, :, None]
        fq = fq[:, :, None]
    dxyz = dxyz[0]
    M = dxyz.shape[0]
    dxyz = F.relu(linear_bn(dxyz, P.sub('linear_p.0')))
    dxyz = F.linear(dxyz, P['linear_p.2.weight'], P['linear_p.2.bias'])                      # [M, K, out]
    mid, out = fk.shape[-1], fv.shape[-1]
    w = fk - fq[0] + dxyz.view(M, K, out // mid, mid).sum(2)
    w = F.relu(batchnorm_lastdim(w, P.sub('bn_w')))            # BatchNorm1d on [M, mid, K] == per channel over (M, K)
    w = F.relu(linear_bn(w, P.sub('linear_w.1')))
    w = F.linear(w, P['linear_w.3.weight'], P['linear_w.3.bias'])                             # [M, K, out / s]
    w = torch.softmax(w, dim=1)
    s = share_planes
    new = ((fv + dxyz).view(M, K, s, out // s) * w.unsqueeze(2)).sum(1).view(M, out)
    short = feats if sparse_xyz is None else gather_rows(feats, idx).max(2)[0]
    if P.has('unary_shortcut.mlp.c.weight'):
        short = unary_block(short, P.sub('unary_shortcut'), relu=False)
    return F.leaky_relu(new + short, 0.1)


def _fused_linear(P: Params, agg):
    """Linear after the aggregate: PConvLinearOpt(+BatchNorm1d) when PCONV_OPT, else
    Linear_BN / nn.Linear.   (layers.py:591-602,698-719)"""
    if P.has('pconv_linear_opt.linear.weight'):
        y = F.linear(agg, P['pconv_linear_opt.linear.weight'], P['pconv_linear_opt.linear.bias'])
        return batchnorm_lastdim(y, P.sub('bn')) if P.has('bn.weight') else y
    return maybe_linear_bn(agg, P.sub('linear'))


def pointconv_layer(P: Params, dense_xyz, dense_feats, idx, dense_norm=None, sparse_xyz=None,
                    sparse_norm=None, use_vi=False, use_pe=False):
    """PointConv.forward.   (layers.py:813-906)"""
    _, wn_in = _geometry(dense_xyz, dense_norm, idx, sparse_xyz, sparse_norm, use_vi)
    w = weightnet(wn_in, P.sub('weightnet'))
    add = wn_in if use_pe else wn_in[..., :0]
    agg = pconv_forward(dense_feats, idx, w, add)
    return F.relu(_fused_linear(P, agg)), wn_in


def pointconv_stride_pe_layer(P: Params, dense_xyz, dense_feats, idx, dense_norm, sparse_xyz=None,
                              sparse_norm=None, vi=None, use_vi=True, drop_scale=None):
    """PointConvStridePE.forward.   (layers.py:631-741; drop_scale as in pcf_layer, :739)"""
    fx = unary_block(dense_feats, P.sub('unary1')) if P.has('unary1.mlp.c.weight') else dense_feats
    rel, wn_in = _geometry(dense_xyz, dense_norm, idx, sparse_xyz, sparse_norm, use_vi, vi)
    pe = weightnet(rel, P.sub('pe_convs'))
    w = weightnet(wn_in, P.sub('weightnet'))
    agg = pconv_forward(fx, idx, w, pe)
    y = F.relu(_fused_linear(P, agg))
    y = unary_block(y, P.sub('unary2'), relu=False)
    short = dense_feats if sparse_xyz is None else gather_rows(dense_feats, idx).max(2)[0]
    if P.has('unary_shortcut.mlp.c.weight'):
        short = unary_block(short, P.sub('unary_shortcut'), relu=False)
    if drop_scale is not None:
        y = y * drop_scale
    return F.leaky_relu(y + short, 0.1), wn_in


def pointconv_transpose_pe_layer(P: Params, sparse_xyz, sparse_feats, idx, sparse_norm, dense_xyz,
                                 dense_norm, dense_feats=None, vi=None, use_vi=True, use_pe=True):
    """PointConvTransposePE.forward (idx indexes the SPARSE level).   (layers.py:1000-1105)"""
    rel = gather_rows(sparse_xyz, idx) - dense_xyz.unsqueeze(2)
    if use_vi:
        wn_in = vi if vi is not None else vi_features(rel, gather_rows(sparse_norm, idx), dense_norm)
    else:
        wn_in = rel
    add = weightnet(rel, P.sub('pe_convs')) if use_pe else rel[..., :0]
    w = weightnet(wn_in, P.sub('weightnet'))
    agg = pconv_forward(sparse_feats, idx, w, add)
    y = F.relu(_fused_linear(P, agg))
    if dense_feats is not None:
        y = y + dense_feats
    i = 0
    while P.has(f'mlp2_convs.{i}.c.weight') or P.has(f'mlp2_convs.{i}.weight'):
        y = F.relu(maybe_linear_bn(y, P.sub(f'mlp2_convs.{i}')))
        i += 1
    return y, wn_in


# --------------------------------------------------------------------------------------
# whole model (model_architecture.py:175-245 backbone forward, :406-502 segmentation forward)
# --------------------------------------------------------------------------------------
def segmentation_model(P: Params, cfg, features, pointclouds, edges_self, edges_forward, edges_propagate, norms,
                       drop_scales=None):
    """PointConvFormer_Segmentation.forward for transformer_type 'PCF' -> logits [1, N0, num_classes].

    cfg: the model config (num_level, guided_level, resblocks, resblocks_back, num_heads, use_level_1, USE_XYZ, USE_VI,
    USE_PE).  Which blocks exist is read from the parameter names, as the reference's constructor decides them
    (:113-165, :376-398).  drop_scales: {block prefix: DropPath factor} for blocks whose residual branch is scaled."""
    use_vi, use_pe, H = cfg.USE_VI is True, bool(cfg.USE_PE), cfg.num_heads
    ds = drop_scales or {}
    B = P.sub('pcf_backbone')

    def block(Q, *args, **kw):
        """A guided (PCFLayer) or unguided (PointConvStridePE) block, by the parameters it holds."""
        if Q.has('guidance_unary.mlp.c.weight'):
            return pcf_layer(Q, *args, num_heads=H, use_vi=use_vi, drop_scale=ds.get(Q.p[:-1]), **kw)
        return pointconv_stride_pe_layer(Q, *args, use_vi=use_vi, drop_scale=ds.get(Q.p[:-1]), **kw)

    x = torch.cat([features, pointclouds[0]], -1) if cfg.USE_XYZ else features
    if cfg.use_level_1:                                                                  # :193-198
        x, vi = pointconv_layer(B.sub('selfpointconv'), pointclouds[0], x, edges_self[0], norms[0], use_vi=use_vi,
                                use_pe=use_pe)
        for name in ('selfpointconv_res1', 'selfpointconv_res2'):
            x, _ = pointconv_stride_pe_layer(B.sub(name), pointclouds[0], x, edges_self[0], norms[0], vi=vi, use_vi=use_vi,
                                             drop_scale=ds.get(B.sub(name).p[:-1]))
    else:                                                                                # :199-202
        x = F.relu(linear_bn(x, B.sub('selfmlp')))
    feats = [x]
    for i in range(cfg.num_level - 1):                                                   # :204-243
        x, _ = block(B.sub(f'pointconv.{i}'), pointclouds[i], feats[-1], edges_forward[i], norms[i], pointclouds[i + 1],
                     norms[i + 1])
        vi, j = None, 0
        while B.has(f'pointconv_res.{i}.{j}.unary2.mlp.c.weight'):
            x, vi_new = block(B.sub(f'pointconv_res.{i}.{j}'), pointclouds[i + 1], x, edges_self[i + 1], norms[i + 1], vi=vi)
            vi = vi_new if vi is None else vi
            j += 1
        feats.append(x)
    x = feats[-1]
    for i in range(cfg.num_level - 1):                                                   # :448-498
        lvl = cfg.num_level - 2 - i
        x, _ = pointconv_transpose_pe_layer(P.sub(f'pointdeconv.{i}'), pointclouds[lvl + 1], x, edges_propagate[lvl],
                                            norms[lvl + 1], pointclouds[lvl], norms[lvl], feats[lvl], use_vi=use_vi,
                                            use_pe=use_pe)
        vi, j = None, 0
        while P.has(f'pointdeconv_res.{i}.{j}.unary2.mlp.c.weight'):
            Q = P.sub(f'pointdeconv_res.{i}.{j}')
            x, vi_new = pointconv_stride_pe_layer(Q, pointclouds[lvl], x, edges_self[lvl], norms[lvl], vi=vi, use_vi=use_vi,
                                                  drop_scale=ds.get(Q.p[:-1]))
            vi = vi_new if vi is None else vi
            j += 1
        feats[lvl] = x
    x = F.relu(linear_bn(x, P.sub('fc1')))                                               # :500-501 (dropout_fc = 0)
    return F.linear(x, P['fc2.weight'], P['fc2.bias'])

