"""Host-side mirror of the reference's operator boundary and of the layers that call it.

Same class names, constructor arguments, ``forward`` signatures, sub-module names (so a reference
``state_dict`` loads unchanged) and config switches (``USE_CUDA_KERNEL``, ``PCONV_OPT``, ``USE_VI``,
``USE_PE``, ``BATCH_NORM``) as

  * ``layer_utils.py``  : ``index_points`` :13-30, ``PConvLinearOpt{,Function}`` :42-86,
                          ``PCF{,Function}`` :89-124, ``PConv{,Function}`` :127-173,
                          ``VI_coordinate_transform`` :176-231, ``Linear_BN`` :241-277, ``UnaryBlock`` :281-319
  * ``layers.py``       : ``MultiHeadGuidance`` :23-68, ``MultiHeadGuidanceQK`` :77-114, ``WeightNet`` :127-191,
                          ``PCFLayer`` :194-416, ``PointTransformerLayer`` :419-539, ``PointConvStridePE`` :542-741,
                          ``PointConv`` :744-906, ``PointConvTransposePE`` :909-1105

but every neighbourhood operation goes to the HIP kernels behind ``pcf_cuda`` / ``pcf_fused``:
this module never gathers with advanced indexing and has no PyTorch fallback for the aggregate --
on a machine without the HIP library the import of ``pcf_cuda`` fails.  Dense per-point linears
and BatchNorm use torch (rocBLAS / MIOpen): they are plain library calls, not the hot path.

Deliberate differences from the reference:
  * ``USE_CUDA_KERNEL`` / ``PCONV_OPT`` are honoured as *which operator entry point* is used
    (separate aggregate + Linear_BN, or the fused aggregate+linear op); both run on HIP.
  * WeightNet is not gradient-checkpointed (288 GB of HBM makes the recompute pointless), so BN
    running statistics are updated once per step, not twice (SURVEY.md section 7 "hard parts").
  * The autograd backward is the true adjoint.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

import pcf_cuda
import pcf_fused


# --------------------------------------------------------------------------------------------------
# operator boundary (layer_utils.py:42-173)
# --------------------------------------------------------------------------------------------------
class PCFFunction(torch.autograd.Function):
    """Guided aggregate.  (layer_utils.py:89-106)  When the caller hands over the inverse CSR of the
    neighbour table, grad_input is a deterministic gather-reduce instead of float atomics."""

    @staticmethod
    def forward(ctx, input_feat, neighbor_inds, guidance, weightnet, inv_neighbors=None, inv_k=None, inv_idx=None):
        out = pcf_cuda.pcf_forward(input_feat, neighbor_inds, guidance, weightnet)
        ctx.save_for_backward(input_feat, neighbor_inds, guidance, weightnet, inv_neighbors, inv_k, inv_idx)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input_feat, neighbor_inds, guidance, weightnet, inv_neighbors, inv_k, inv_idx = ctx.saved_tensors
        if inv_idx is not None:
            gi, gg, gw = pcf_cuda.pcf_backward_csr(grad_output.contiguous(), input_feat, inv_neighbors, inv_k, inv_idx,
                                                   neighbor_inds, guidance, weightnet)
        else:
            gi, gg, gw = pcf_cuda.pcf_backward(grad_output.contiguous(), input_feat, neighbor_inds, guidance, weightnet)
        return gi, None, gg, gw, None, None, None


class PCF(pcf_fused.CounterScope):
    """(layer_utils.py:109-124)"""

    @staticmethod
    def forward(input_features, neighbor_inds, guidance, weightnet, inv_neighbors=None, inv_k=None, inv_idx=None):
        return PCFFunction.apply(input_features, neighbor_inds, guidance, weightnet, inv_neighbors, inv_k, inv_idx)


class PConvFunction(torch.autograd.Function):
    """Unguided aggregate with appended per-edge features.  (layer_utils.py:127-153)"""

    @staticmethod
    def forward(ctx, input_feat, neighbor_inds, weightnet, additional_features):
        out = pcf_cuda.pconv_forward(input_feat, neighbor_inds, weightnet, additional_features)
        ctx.save_for_backward(input_feat, neighbor_inds, weightnet, additional_features)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        gi, gw, ga = pcf_cuda.pconv_backward(grad_output.contiguous(), *ctx.saved_tensors)
        return gi, None, gw, ga


def _empty_additional(input_features, neighbor_inds):
    # the reference builds this on the CPU (layer_utils.py:168), which its own CHECK_CUDA rejects
    B, Nout, K = neighbor_inds.shape
    return input_features.new_zeros(B, Nout, K, 0)


class PConv(pcf_fused.CounterScope):
    """(layer_utils.py:156-173)"""

    @staticmethod
    def forward(input_features, neighbor_inds, weightnet, additional_features=None):
        if additional_features is None:
            additional_features = _empty_additional(input_features, neighbor_inds)
        return PConvFunction.apply(input_features, neighbor_inds, weightnet, additional_features)


class PConvLinearOptFunction(torch.autograd.Function):
    """Aggregate + linear, backward over the inverse CSR.  (layer_utils.py:42-70)"""

    @staticmethod
    def forward(ctx, input_feat, neighbor_inds, inverse_neighbors, inverse_k, inverse_idx, weightnet,
                additional_features, linear_weights, linear_bias):
        output, pconv_output = pcf_cuda.pconv_linear_cutlass_forward(
            input_feat, neighbor_inds, weightnet, additional_features, linear_weights, linear_bias)
        ctx.save_for_backward(input_feat, inverse_neighbors, inverse_k, inverse_idx, neighbor_inds, weightnet,
                              additional_features, linear_weights, pconv_output)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        (input_feat, inverse_neighbors, inverse_k, inverse_idx, neighbor_inds, weightnet, additional_features,
         linear_weights, pconv_output) = ctx.saved_tensors
        g = pcf_cuda.pconv_linear_opt_backward(grad_output.contiguous(), input_feat, inverse_neighbors, inverse_k,
                                               inverse_idx, neighbor_inds, weightnet, additional_features,
                                               linear_weights, pconv_output)
        return g[0], None, None, None, None, g[1], g[2], g[3], g[4]


class PConvLinearOpt(pcf_fused.CounterScope):
    """(layer_utils.py:73-86)"""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)

    def forward(self, input_features, neighbor_inds, inverse_neighbors, inverse_k, inverse_idx, weightnet,
                additional_features=None):
        if additional_features is None:
            additional_features = _empty_additional(input_features, neighbor_inds)
        if inverse_neighbors is None:      # the reference would crash; build the CSR on the fly instead
            inverse_neighbors, inverse_k, inverse_idx = pcf_cuda.compute_knn_inverse(
                neighbor_inds, input_features.shape[1])
        return PConvLinearOptFunction.apply(input_features, neighbor_inds, inverse_neighbors, inverse_k, inverse_idx,
                                            weightnet, additional_features, self.linear.weight, self.linear.bias)


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,K] -> [B,S,(K,)C] on the HIP row-gather (differentiable).
    (layer_utils.py:13-30)"""
    return pcf_fused.gather_rows(points, idx)


def VI_coordinate_transform(localized_xyz, gathered_norm, sparse_xyz_norm, K=None):
    """Viewpoint-invariant descriptor from already-gathered tensors (layer_utils.py:176-231).  The
    layers below call the fused gather+transform kernel instead; this entry point exists for callers
    that hold the gathered tensors."""
    return pcf_fused.vi_from_gathered(localized_xyz, gathered_norm, sparse_xyz_norm)


# --------------------------------------------------------------------------------------------------
# Linear_BN / UnaryBlock (layer_utils.py:241-319)
# --------------------------------------------------------------------------------------------------
class Linear_BN(pcf_fused.CounterScope):
    """Linear followed by BatchNorm over every axis but the last.  ``bn_ver`` is accepted for
    signature compatibility; both versions normalise the channel (last) axis."""

    def __init__(self, in_dim, out_dim, bn_ver='2d', bn_weight_init=1, bn_momentum=0.1):
        super().__init__()
        self.c = nn.Linear(in_dim, out_dim)
        self.bn_ver = bn_ver
        self.bn = nn.BatchNorm1d(out_dim, momentum=bn_momentum)
        nn.init.constant_(self.bn.weight, bn_weight_init)

    @torch.no_grad()
    def fuse(self):
        """Fold the running statistics into one nn.Linear for inference.  (layer_utils.py:260-270)"""
        s = self.bn.weight * torch.rsqrt(self.bn.running_var + self.bn.eps)
        lin = nn.Linear(self.c.in_features, self.c.out_features).to(self.c.weight.device)
        lin.weight.copy_(self.c.weight * s[:, None])
        lin.bias.copy_(self.bn.bias + (self.c.bias - self.bn.running_mean) * s)
        return lin

    def forward(self, x, act=pcf_fused.ACT_NONE):
        """BN(x W^T + b), optionally followed by `act` (a pcf_fused.ACT_* code) in the same kernel."""
        if pcf_fused.rowlin_supported(self.c.in_features, self.c.out_features):
            return pcf_fused.linear_bn_act(x, self.c.weight, self.c.bias, self.bn, act, self.training)
        # wide point-level layers: fp32 MFMA contraction + column-wise BatchNorm kernels
        return pcf_fused.wide_linear_bn_act(x, self.c.weight, self.c.bias, self.bn, act, self.training)

    def forward_residual(self, x, residual, act):
        """act(BN(x W^T + b) + residual): in one kernel after the contraction on the wide path."""
        if pcf_fused.rowlin_supported(self.c.in_features, self.c.out_features):
            return _apply_act(self.forward(x) + residual, act)
        return pcf_fused.wide_linear_bn_act(x, self.c.weight, self.c.bias, self.bn, act, self.training, residual=residual)


def _apply_act(y, act):
    if act == pcf_fused.ACT_RELU:
        return F.relu(y)
    if act == pcf_fused.ACT_LEAKY:
        return F.leaky_relu(y, 0.1)
    if act == pcf_fused.ACT_SIGMOID:
        return torch.sigmoid(y)
    return y


def _linear_act(layer, x, act):
    """Linear_BN or plain nn.Linear (cfg.BATCH_NORM False) followed by `act`, fused when narrow."""
    if isinstance(layer, Linear_BN):
        return layer(x, act)
    if pcf_fused.rowlin_supported(layer.in_features, layer.out_features):
        return pcf_fused.linear_bn_act(x, layer.weight, layer.bias, None, act, layer.training)
    return pcf_fused.wide_linear_bn_act(x, layer.weight, layer.bias, None, act, layer.training)


class UnaryBlock(pcf_fused.CounterScope):
    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super().__init__()
        self.in_dim, self.out_dim, self.use_bn, self.no_relu = in_dim, out_dim, use_bn, no_relu
        self.mlp = Linear_BN(in_dim, out_dim, bn_momentum=bn_momentum, bn_ver='1d') if use_bn \
            else nn.Linear(in_dim, out_dim)

    def forward(self, x):
        return _linear_act(self.mlp, x, pcf_fused.ACT_NONE if self.no_relu else pcf_fused.ACT_LEAKY)

    def forward_residual(self, x, residual, act):
        """act(block(x) + residual) for a block without its own activation (the tail of the residual layers)."""
        if self.no_relu and isinstance(self.mlp, Linear_BN):
            return self.mlp.forward_residual(x, residual, act)
        return _apply_act(self.forward(x) + residual, act)


# --------------------------------------------------------------------------------------------------
# per-edge MLPs (layers.py:23-68, 127-191)
# --------------------------------------------------------------------------------------------------
class MultiHeadGuidance(pcf_fused.CounterScope):
    """sigmoid(MLP_{C -> 8 -> heads}(query - key)), ReLU between the two layers."""

    def __init__(self, cfg, num_heads: int, num_hiddens: int):
        super().__init__()
        self.num_heads, self.dim = num_heads, num_hiddens
        ln = bool(getattr(cfg, 'layer_norm_guidance', False))       # ablation switch, False in every BASELINE config
        self.layer_norm_q = nn.LayerNorm(num_hiddens) if ln else nn.Identity()
        self.layer_norm_k = nn.LayerNorm(num_hiddens) if ln else nn.Identity()
        dims = [num_hiddens, 8, num_heads]
        self.mlp = nn.ModuleList(
            Linear_BN(a, b) if cfg.BATCH_NORM else nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    @property
    def layer_norm(self):
        return isinstance(self.layer_norm_q, nn.LayerNorm)

    def forward(self, guidance_query, guidance_key):
        if self.layer_norm:        # LayerNorm on the query and on the key (layers.py:52-53), on HIP
            guidance_query = pcf_fused.layer_norm(guidance_query, self.layer_norm_q)
            guidance_key = pcf_fused.layer_norm(guidance_key, self.layer_norm_k)
        return self.forward_diff(guidance_query - guidance_key)

    def forward_split(self, guidance_x, nei_inds, feat_pe):
        """Scores for SELF neighbourhoods (key = neighbour 0) without forming the query tensor.

        With the first layer's weight split as W = [Wa | Wb] over the gathered / positional halves,
        W.(q - key) = (u[idx] + Wb.pe) - (same for the key edge) where u = Wa.guidance_x is a per-point
        product: an 8-wide gather replaces the [B,M,K,2G] query, key and difference tensors."""
        first = self.mlp[0]
        lin = first.c if isinstance(first, Linear_BN) else first
        bn = first.bn if isinstance(first, Linear_BN) else None
        G = guidance_x.shape[-1]
        W = lin.weight
        u = pcf_fused.linear_bn_act(guidance_x, W[:, :G], W.new_zeros(W.shape[0]), None, pcf_fused.ACT_NONE, self.training)
        last = len(self.mlp) - 1
        s = pcf_fused.linear_bn_act(feat_pe, W[:, G:], lin.bias, bn,
                                    pcf_fused.ACT_SIGMOID if last == 0 else pcf_fused.ACT_RELU, self.training,
                                    gadd=u, gidx=nei_inds, group=nei_inds.shape[2])
        for i in range(1, len(self.mlp)):
            s = _linear_act(self.mlp[i], s, pcf_fused.ACT_SIGMOID if i == last else pcf_fused.ACT_RELU)
        return s

    def forward_diff(self, s):
        """Scores from the already-formed difference query - key."""
        last = len(self.mlp) - 1
        for i, layer in enumerate(self.mlp):
            s = _linear_act(layer, s, pcf_fused.ACT_SIGMOID if i == last else pcf_fused.ACT_RELU)
        return s


class MultiHeadGuidanceQK(pcf_fused.CounterScope):
    """sigmoid(scale * <W q, W key>) per head -- the inner-product form of the guidance, an ablation no BASELINE config
    uses (layers.py:77-114; selected by cfg.attention_type != 'subtraction', :264-269).  The same Linear_BN runs over
    the query tensor and over the key tensor repeated K times (two BatchNorm calls, each with its own batch statistics,
    exactly as upstream); both go through the contraction + column-BatchNorm kernels, the per-head dot product and
    the sigmoid are one HIP kernel (csrc/attention_ops.hip)."""

    def __init__(self, cfg, num_heads: int, num_hiddens: int, key_dim: int):
        super().__init__()
        assert num_hiddens % num_heads == 0, 'num_hiddens: %d, num_heads: %d' % (num_hiddens, num_heads)
        self.cfg, self.dim, self.num_heads, self.key_dim = cfg, num_hiddens, num_heads, key_dim
        self.scale = key_dim ** -0.5
        self.qk_linear = Linear_BN(self.dim, key_dim * num_heads)

    def forward(self, q, k):
        """q [B,N,K,C]; k [B,N,K,C] or [B,N,1,C] (one centre row per neighbourhood) -> scores [B,N,K,heads]."""
        B, N, K, _ = q.shape
        if k.shape[2] != K:
            k = k.expand(-1, -1, K, -1)
        qq = self.qk_linear(q.contiguous()).view(B, N, K, self.num_heads, -1)
        kk = self.qk_linear(k.contiguous()).view(B, N, K, self.num_heads, -1)[:, :, 0]
        return pcf_fused.qk_score(qq, kk, self.scale)          # per-head dot product + sigmoid in one kernel


class WeightNet(pcf_fused.CounterScope):
    """Linear_BN + ReLU after every layer, the last one included."""

    def __init__(self, in_channel, out_channel, hidden_unit=(8, 8), efficient=False):
        super().__init__()
        dims = [in_channel] + list(hidden_unit or []) + [out_channel]
        self.mlp_convs = nn.ModuleList(Linear_BN(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.efficient = efficient   # accepted, unused: no checkpointing on MI355X

    def forward(self, localized_xyz):
        w = localized_xyz
        convs = list(self.mlp_convs)
        if len(convs) == 3 and not w.requires_grad and not getattr(self, 'no_chain', False) \
                and (self.training or not torch.is_grad_enabled()) \
                and pcf_fused.same_bn_hyperparameters([m.bn for m in convs]) \
                and pcf_fused.weightnet_chain_supported(convs[0].c.in_features, (convs[0].c.out_features, convs[1].c.out_features),
                                                        convs[2].c.out_features, w.numel() // max(1, w.shape[-1])):
            # the three layers in one fused chain (four passes forward, three backward; csrc/edge_chain*.hip)
            return pcf_fused.weightnet_chain(w, [(m.c, m.bn) for m in convs], self.training)
        if len(convs) == 2 and self.training and w.numel() > 0 and not w.requires_grad and not getattr(self, 'no_chain', False) \
                and pcf_fused.point_chain_ok(convs[0].bn, convs[1].bn) and convs[0].bn.eps == convs[1].bn.eps \
                and pcf_fused.pe_chain_supported(convs[0].c.in_features, (convs[0].c.out_features,), convs[1].c.out_features):
            # pe_convs (3 -> out/4 -> min(out/4, 32)): both layers per pass, nothing but the output stored
            return pcf_fused.pe_chain(w, [(m.c, m.bn) for m in convs])
        for conv in convs:
            w = conv(w, pcf_fused.ACT_RELU)
        return w


# --------------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------------
def _edge_geometry(cfg_use_vi, ref_xyz, ref_norm, nei_inds, ctr_xyz, ctr_norm, vi_features, want_rel=True):
    """-> (localized_xyz or None, weightNetInput).  One fused HIP kernel gathers the neighbour
    coordinates / normals and emits the offsets (unless want_rel is False and the VI descriptor is produced)
    and the 12-channel VI descriptor."""
    if cfg_use_vi and vi_features is not None:
        return None, vi_features
    rel, vi = pcf_fused.edge_geometry(ref_xyz, ref_norm if cfg_use_vi else None, nei_inds, ctr_xyz,
                                      ctr_norm if cfg_use_vi else None, want_rel=want_rel)
    return rel, (vi if cfg_use_vi else rel)


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath, which layers.py:9 imports and :237-238, :571-572,
    :951-952 instantiate): in training the residual branch of a sample is dropped with probability `drop_prob` and
    scaled by 1/keep otherwise; identity in eval.  One Bernoulli draw per batch row -- with the packed B = 1 layout
    (layers.py:216) that is one draw per block call.  The draw stays on the device (no host sync, capturable)."""

    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = float(drop_prob), scale_by_keep

    def draw(self, x):
        """-> the per-sample factor [B, 1, ...] (0 or 1/keep), or None when the module is the identity."""
        if self.drop_prob == 0. or not self.training:
            return None
        keep = 1. - self.drop_prob
        m = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        if keep > 0. and self.scale_by_keep:
            m.div_(keep)
        return m

    def forward(self, x):
        m = self.draw(x)
        return x if m is None else x * m

    def extra_repr(self):
        return f'drop_prob={round(self.drop_prob, 3):0.3f}'


def _drop_path(cfg):
    rate = float(getattr(cfg, 'drop_path_rate', 0.))
    return DropPath(rate) if rate > 0. else nn.Identity()


def _residual_tail(block, drop_path, x, shortcut, act):
    """act(drop_path(block(x)) + shortcut)  (layers.py:414, :739).  Without a drop the sum and the activation ride in
    the block's BatchNorm kernel; with one the branch is scaled by the drawn per-sample factor first."""
    m = drop_path.draw(x) if isinstance(drop_path, DropPath) else None
    if m is None:
        return block.forward_residual(x, shortcut, act)
    return _apply_act(block(x) * m + shortcut, act)


class PCFLayer(pcf_fused.CounterScope):
    """PointConvFormer layer.  (layers.py:194-416)"""

    def __init__(self, in_channel, out_channel, cfg, weightnet=[9, 16], num_heads=4, guidance_feat_len=32):
        super().__init__()
        self.cfg, self.in_channel, self.out_channel, self.num_heads = cfg, in_channel, out_channel, num_heads
        self.drop_path = _drop_path(cfg)
        self._zero8 = torch.zeros(8)        # bias of the per-point half of the first guidance layer (not a parameter / buffer)
        self.mlp_conv = Linear_BN(12, guidance_feat_len) if cfg.BATCH_NORM else nn.Linear(12, guidance_feat_len)
        mid = out_channel // 4
        self.unary1 = UnaryBlock(in_channel, mid, use_bn=True, bn_momentum=0.1) if in_channel != mid else nn.Identity()
        self.guidance_unary = UnaryBlock(mid, guidance_feat_len, use_bn=True, bn_momentum=0.1, no_relu=True)
        assert (out_channel // 2) % num_heads == 0
        self.guidance_weight = MultiHeadGuidance(cfg, num_heads, 2 * guidance_feat_len) \
            if cfg.attention_type == 'subtraction' else MultiHeadGuidanceQK(cfg, num_heads, 2 * guidance_feat_len, key_dim=16)
        self.weightnet = WeightNet(weightnet[0], weightnet[1], efficient=True)
        self.linear = Linear_BN(mid * weightnet[-1], out_channel // 2, bn_ver='1d') if cfg.BATCH_NORM \
            else nn.Linear(mid * weightnet[-1], out_channel // 2)
        self.dropout = nn.Dropout(p=cfg.dropout_rate) if cfg.dropout_rate > 0. else nn.Identity()
        self.unary2 = UnaryBlock(out_channel // 2, out_channel, use_bn=True, bn_momentum=0.1, no_relu=True)
        self.unary_shortcut = UnaryBlock(in_channel, out_channel, use_bn=True, bn_momentum=0.1, no_relu=True) \
            if in_channel != out_channel else nn.Identity()

    def _chain_layers(self, wn_in, nei_inds):
        """The six (Linear, BatchNorm) pairs of the edge graph if the fused chain kernel covers this layer
        (csrc/edge_chain.hip), else None."""
        if getattr(self.cfg, 'NO_EDGE_CHAIN', False) or getattr(self.cfg, 'DETERMINISTIC_BACKWARD', False) \
                or not isinstance(self.guidance_weight, MultiHeadGuidance) or self.guidance_weight.layer_norm:
            return None
        # the chain's backward restarts from batch statistics and returns no gradient for the VI descriptor: eval-mode
        # forward under autograd (frozen-BN fine-tuning, saliency) and a differentiable descriptor go layer by layer
        if (not self.training and torch.is_grad_enabled()) or wn_in.requires_grad:
            return None
        gw, wn = self.guidance_weight.mlp, self.weightnet.mlp_convs
        mods = [self.mlp_conv] + list(gw) + list(wn)
        if len(gw) != 2 or len(wn) != 3 or not all(isinstance(m, Linear_BN) for m in mods):
            return None
        hidden_ok = gw[0].c.out_features == 8 and wn[0].c.out_features == 8 and wn[1].c.out_features == 8 \
            and gw[0].c.in_features == 2 * self.mlp_conv.c.out_features
        if not pcf_fused.pcf_chain_supported(wn_in.shape[-1], self.mlp_conv.c.out_features, gw[1].c.out_features,
                                             wn[2].c.out_features, nei_inds.shape[2], hidden_ok, nei_inds.numel(),
                                             nei_inds.shape[1] * nei_inds.shape[2]):
            return None
        if not pcf_fused.same_bn_hyperparameters([m.bn for m in mods]):
            return None
        return [(m.c, m.bn) for m in mods]

    def forward(self, dense_xyz, dense_feats, nei_inds, dense_xyz_norm, sparse_xyz=None, sparse_xyz_norm=None,
                vi_features=None, inv_neighbors=None, inv_k=None, inv_idx=None):
        strided = sparse_xyz is not None
        ctr_xyz = sparse_xyz if strided else dense_xyz
        ctr_norm = sparse_xyz_norm if strided else dense_xyz_norm
        nei_inds = nei_inds.contiguous()
        _, wn_in = _edge_geometry(self.cfg.USE_VI is True, dense_xyz, dense_xyz_norm, nei_inds, ctr_xyz, ctr_norm,
                                  vi_features, want_rel=False)
        chain = self._chain_layers(wn_in, nei_inds)
        if strided and chain is not None and (nei_inds.shape[2] < 2 or getattr(self.cfg, 'EDGE_CHAIN_LAYERWISE_BACKWARD', False)):
            chain = None            # the maximum-key form of the chain: K >= 2, fused backward only
        fused_points = self.training and not getattr(self.cfg, 'NO_POINT_CHAIN', False)
        force_flin = bool(getattr(self.cfg, 'FLIN_POINT_CHAINS', False))     # fused contraction chains whatever the row count
        if chain is not None:
            # BatchNorm everywhere: the whole edge graph in four fused passes forward, three backward (self neighbourhoods:
            # key = neighbour 0; strided: key = maximum over the neighbourhood)
            g1 = self.guidance_weight.mlp[0].c
            G = self.guidance_unary.out_dim
            Wa, Wb = pcf_fused.split_columns(g1.weight, G)      # gathered half | positional half
            u1 = self.unary1 if isinstance(self.unary1, UnaryBlock) else None
            ukey = None
            if strided:
                # key = max of the query over the neighbourhood (layers.py:372-375): its gathered half per centre, Wa . max_k
                # guidance_x[idx]; the positional half is taken inside the chain kernels
                feats_x = self.unary1(dense_feats)
                guidance_x = self.guidance_unary(feats_x).contiguous()
                if self._zero8.device != Wa.device:
                    self._zero8 = self._zero8.to(Wa.device)
                zero = self._zero8[:g1.out_features]
                u = pcf_fused.linear_bn_act(guidance_x, Wa, zero, None, pcf_fused.ACT_NONE, self.training)
                ukey = pcf_fused.linear_bn_act(pcf_fused.gather_max(guidance_x, nei_inds), Wa, zero, None, pcf_fused.ACT_NONE,
                                               self.training)
            elif fused_points and pcf_fused.head_chain_pays(dense_feats, u1, self.guidance_unary, force=force_flin) \
                    and pcf_fused.point_chain_ok(self.guidance_unary.mlp.bn, *([u1.mlp.bn] if u1 is not None else [])):
                # unary1 -> guidance_unary -> u in three launches (BatchNorms folded into the contractions)
                feats_x, u = pcf_fused.point_head(dense_feats, u1, self.guidance_unary, Wa)
            else:
                feats_x = self.unary1(dense_feats)
                guidance_x = self.guidance_unary(feats_x)
                if self._zero8.device != Wa.device:
                    self._zero8 = self._zero8.to(Wa.device)
                u = pcf_fused.linear_bn_act(guidance_x, Wa, self._zero8[:g1.out_features], None, pcf_fused.ACT_NONE, self.training)
            agg = pcf_fused.pcf_chain(wn_in.contiguous(), nei_inds, u, feats_x.contiguous(), chain, self.training,
                                      fused_backward=not getattr(self.cfg, 'EDGE_CHAIN_LAYERWISE_BACKWARD', False),
                                      g1_positional_weight=Wb, ukey=ukey)
        else:
            feats_x = self.unary1(dense_feats)
            guidance_x = self.guidance_unary(feats_x)
            feat_pe = _linear_act(self.mlp_conv, wn_in, pcf_fused.ACT_RELU)
            if isinstance(self.guidance_weight, MultiHeadGuidanceQK) or self.guidance_weight.layer_norm:
                # ablations (layers.py:370-381): the query tensor is formed, the key is its centre row / maximum over K
                query = torch.cat([pcf_fused.gather_rows(guidance_x.contiguous(), nei_inds), feat_pe], -1)
                key = query.max(dim=2, keepdim=True)[0] if strided else query[:, :, :1]
                guidance_score = self.guidance_weight(query, key)
            elif not strided and pcf_fused.split_guidance_supported(nei_inds.shape[2], 8) \
                    and pcf_fused.rowlin_supported(feat_pe.shape[-1], 8):
                # self neighbourhoods: key = neighbour 0; the first guidance layer absorbs the q - key algebra
                guidance_score = self.guidance_weight.forward_split(guidance_x.contiguous(), nei_inds, feat_pe)
            else:
                # strided: key = max over the neighbourhood, formed explicitly in one kernel
                diff = pcf_fused.guidance_diff(guidance_x.contiguous(), nei_inds, feat_pe, use_max=strided)
                guidance_score = self.guidance_weight.forward_diff(diff)
            weights = self.weightnet(wn_in)
            if not getattr(self.cfg, 'DETERMINISTIC_BACKWARD', False):
                inv_neighbors = inv_k = inv_idx = None      # float atomics are ~25 % faster than the CSR reduce here
            agg = PCF.forward(feats_x.contiguous(), nei_inds, guidance_score.contiguous(), weights.contiguous(),
                              inv_neighbors, inv_k, inv_idx)
        sparse_feats = pcf_fused.gather_max(dense_feats, nei_inds) if strided else dense_feats
        shortcut = self.unary_shortcut(sparse_feats)
        if fused_points and isinstance(self.linear, Linear_BN) and isinstance(self.dropout, nn.Identity) \
                and pcf_fused.tail_chain_pays(agg, self.linear, self.unary2, force=force_flin) \
                and not (isinstance(self.drop_path, DropPath) and self.drop_path.drop_prob > 0.) \
                and pcf_fused.point_chain_ok(self.linear.bn, self.unary2.mlp.bn):
            # linear + ReLU -> unary2 -> + shortcut -> LeakyReLU in three launches forward, five backward
            return pcf_fused.point_tail(agg, shortcut, self.linear, self.unary2), wn_in
        # leaky_relu(drop_path(unary2(.)) + shortcut)   (layers.py:397-414; drop_path_rate 0.2 in configPCF_2cm_PTF2)
        new_feat = _residual_tail(self.unary2, self.drop_path, self.dropout(_linear_act(self.linear, agg, pcf_fused.ACT_RELU)),
                                  shortcut, pcf_fused.ACT_LEAKY)
        return new_feat, wn_in


class PointTransformerLayer(pcf_fused.CounterScope):
    """PointTransformer block, the reference's ablation against PCFLayer (layers.py:419-539; selected by
    cfg.transformer_type != 'PCF', model_architecture.py:138-176): vector attention over the K neighbours with a
    softmax, positional term from the coordinate offsets.  Same sub-module names as upstream (state_dicts load
    unchanged).  Linears, BatchNorms, gathers and the coordinate offsets run on the HIP kernels of the hot path, the
    softmax over K and the weighted neighbour sum on csrc/attention_ops.hip (this block is not on any BASELINE config's path)."""

    def __init__(self, in_planes, out_planes, share_planes=8):
        super().__init__()
        self.mid_planes = mid_planes = out_planes // 1
        self.out_planes, self.share_planes = out_planes, share_planes
        self.linear_q = nn.Linear(in_planes, mid_planes)
        self.linear_k = nn.Linear(in_planes, mid_planes)
        self.linear_v = nn.Linear(in_planes, out_planes)
        self.linear_p = nn.Sequential(Linear_BN(3, 3, bn_ver='1d'), nn.ReLU(inplace=True), nn.Linear(3, out_planes))
        self.bn_w = nn.BatchNorm1d(mid_planes)
        self.linear_w = nn.Sequential(nn.ReLU(inplace=True), Linear_BN(mid_planes, mid_planes // share_planes, bn_ver='1d'),
                                      nn.ReLU(inplace=True), nn.Linear(mid_planes // share_planes, out_planes // share_planes))
        self.unary_shortcut = UnaryBlock(in_planes, out_planes, use_bn=True, bn_momentum=0.1, no_relu=True) \
            if in_planes != out_planes else nn.Identity()

    def forward(self, xyz, feats, nei_ind, sparse_xyz=None):
        strided = sparse_xyz is not None
        nei_ind = nei_ind.contiguous()
        B, M, K = nei_ind.shape
        feats = feats.contiguous()
        feats_q = _linear_act(self.linear_q, feats, pcf_fused.ACT_NONE)
        feats_k = pcf_fused.gather_rows(_linear_act(self.linear_k, feats, pcf_fused.ACT_NONE), nei_ind)     # [B,M,K,mid]
        feats_v = pcf_fused.gather_rows(_linear_act(self.linear_v, feats, pcf_fused.ACT_NONE), nei_ind)     # [B,M,K,out]
        if strided:
            feats_q = pcf_fused.gather_rows(feats_q, nei_ind[:, :, :1].contiguous())                        # [B,M,1,mid]
        else:
            feats_q = feats_q[:, :, None]
        dxyz, _ = pcf_fused.edge_geometry(xyz, None, nei_ind, sparse_xyz if strided else xyz, None)           # xyz[idx] - centre
        dxyz = _linear_act(self.linear_p[2], self.linear_p[0](dxyz, pcf_fused.ACT_RELU), pcf_fused.ACT_NONE)  # [B,M,K,out]
        w = feats_k - feats_q + dxyz.view(B, M, K, self.out_planes // self.mid_planes, self.mid_planes).sum(3)
        w = pcf_fused.bn_act(w, self.bn_w, pcf_fused.ACT_RELU, self.training)        # BatchNorm1d over (M, K) + the first ReLU
        w = _linear_act(self.linear_w[3], self.linear_w[1](w, pcf_fused.ACT_RELU), pcf_fused.ACT_NONE)
        # softmax over the K neighbours and the weighted neighbour sum (share_planes groups share a weight) in one kernel
        new_feats = pcf_fused.softmax_aggregate(feats_v + dxyz, w)
        sparse_feats = pcf_fused.gather_max(feats, nei_ind) if strided else feats
        return F.leaky_relu(new_feats + self.unary_shortcut(sparse_feats), 0.1)


class _ConvTail(pcf_fused.CounterScope):
    """Shared tail of the PointConv family: aggregate + linear, chosen by PCONV_OPT / USE_CUDA_KERNEL."""

    def _build_linear(self, cfg, in_features, out_features):
        if cfg.PCONV_OPT:
            self.pconv_linear_opt = PConvLinearOpt(in_features, out_features)
            if cfg.BATCH_NORM:
                self.bn = nn.BatchNorm1d(out_features, momentum=0.1)
        else:
            self.linear = Linear_BN(in_features, out_features, bn_ver='1d') if cfg.BATCH_NORM \
                else nn.Linear(in_features, out_features)

    def _aggregate_linear(self, feats, nei_inds, weights, additional, inv_neighbors, inv_k, inv_idx):
        """relu(BN(linear(aggregate))): the ReLU every caller applies (layers.py:721, 901, 1094) rides in the
        BatchNorm kernel."""
        feats, weights = feats.contiguous(), weights.contiguous()
        additional = None if additional is None else additional.contiguous()
        if self.cfg.PCONV_OPT:
            y = self.pconv_linear_opt(feats, nei_inds, inv_neighbors, inv_k, inv_idx, weights, additional)
            if self.cfg.BATCH_NORM:
                return pcf_fused.bn_act(y, self.bn, pcf_fused.ACT_RELU, self.training)
            return F.relu(y)
        return _linear_act(self.linear, PConv.forward(feats, nei_inds, weights, additional), pcf_fused.ACT_RELU)


class PointConvStridePE(_ConvTail):
    """PointConv with a learned positional embedding appended to the gathered features.  (layers.py:542-741)"""

    def __init__(self, in_channel, out_channel, cfg, weightnet=[9, 16]):
        super().__init__()
        self.cfg, self.in_channel, self.out_channel = cfg, in_channel, out_channel
        self.drop_path = _drop_path(cfg)
        mid = out_channel // 4
        last_ch = min(mid, 32)
        self.pe_convs = WeightNet(3, last_ch, hidden_unit=[mid], efficient=True)
        self.unary1 = UnaryBlock(in_channel, mid, use_bn=True, bn_momentum=0.1) if in_channel != mid else nn.Identity()
        self.weightnet = WeightNet(weightnet[0], weightnet[1], efficient=True)
        self._build_linear(cfg, (mid + last_ch) * weightnet[-1], out_channel // 2)
        self.dropout = nn.Dropout(p=cfg.dropout_rate) if cfg.dropout_rate > 0. else nn.Identity()
        self.unary2 = UnaryBlock(out_channel // 2, out_channel, use_bn=True, bn_momentum=0.1, no_relu=True)
        self.unary_shortcut = UnaryBlock(in_channel, out_channel, use_bn=True, bn_momentum=0.1, no_relu=True) \
            if in_channel != out_channel else nn.Identity()

    def forward(self, dense_xyz, dense_feats, nei_inds, dense_xyz_norm, sparse_xyz=None, sparse_xyz_norm=None,
                vi_features=None, inv_neighbors=None, inv_k=None, inv_idx=None):
        strided = sparse_xyz is not None
        ctr_xyz = sparse_xyz if strided else dense_xyz
        ctr_norm = sparse_xyz_norm if strided else dense_xyz_norm
        nei_inds = nei_inds.contiguous()
        feats_x = self.unary1(dense_feats)
        rel, wn_in = _edge_geometry(self.cfg.USE_VI is True, dense_xyz, dense_xyz_norm, nei_inds, ctr_xyz, ctr_norm,
                                    vi_features)
        if rel is None:                       # VI features were handed in: offsets still needed for the PE
            rel, _ = pcf_fused.edge_geometry(dense_xyz, None, nei_inds, ctr_xyz, None)
        feat_pe = self.pe_convs(rel)
        weights = self.weightnet(wn_in)
        y = self._aggregate_linear(feats_x, nei_inds, weights, feat_pe, inv_neighbors, inv_k, inv_idx)
        sparse_feats = pcf_fused.gather_max(dense_feats, nei_inds) if strided else dense_feats
        shortcut = self.unary_shortcut(sparse_feats)
        return _residual_tail(self.unary2, self.drop_path, self.dropout(y), shortcut, pcf_fused.ACT_LEAKY), wn_in


class PointConv(_ConvTail):
    """First-layer (VI_)PointConv without bottleneck.  (layers.py:744-906)"""

    def __init__(self, in_channel, out_channel, cfg, weightnet=[9, 16], USE_VI=None):
        super().__init__()
        self.cfg, self.in_channel, self.out_channel = cfg, in_channel, out_channel
        self.USE_VI = cfg.USE_VI if USE_VI is None else USE_VI
        last_ch = in_channel + ((12 if self.USE_VI else 3) if cfg.USE_PE else 0)
        self.weightnet = WeightNet(weightnet[0], weightnet[1], efficient=True)
        self._build_linear(cfg, last_ch * weightnet[-1], out_channel)
        self.dropout = nn.Dropout(p=cfg.dropout_rate) if cfg.dropout_rate > 0. else nn.Identity()

    def forward(self, dense_xyz, dense_feats, nei_inds, dense_xyz_norm=None, sparse_xyz=None, sparse_xyz_norm=None,
                inv_neighbors=None, inv_k=None, inv_idx=None):
        strided = sparse_xyz is not None
        nei_inds = nei_inds.contiguous()
        _, wn_in = _edge_geometry(self.USE_VI is True, dense_xyz, dense_xyz_norm, nei_inds,
                                  sparse_xyz if strided else dense_xyz,
                                  sparse_xyz_norm if strided else dense_xyz_norm, None)
        weights = self.weightnet(wn_in)
        additional = wn_in if self.cfg.USE_PE else None
        y = self._aggregate_linear(dense_feats, nei_inds, weights, additional, inv_neighbors, inv_k, inv_idx)
        return self.dropout(y), wn_in


class PointConvTransposePE(_ConvTail):
    """Upsampling PointConv: features move from the sparse level to the dense one.  (layers.py:909-1105)"""

    def __init__(self, in_channel, out_channel, cfg, weightnet=[9, 16], mlp2=None):
        super().__init__()
        self.cfg, self.in_channel, self.out_channel = cfg, in_channel, out_channel
        self.drop_path = _drop_path(cfg)
        if cfg.USE_PE:
            last_ch = min(out_channel // 4, 32)
            self.pe_convs = WeightNet(3, last_ch, hidden_unit=[out_channel // 4], efficient=True)
        else:
            last_ch = 0
            self.pe_convs = nn.ModuleList()
        self.weightnet = WeightNet(weightnet[0], weightnet[1], efficient=True)
        self._build_linear(cfg, (last_ch + in_channel) * weightnet[-1], out_channel)
        self.dropout = nn.Dropout(p=cfg.dropout_rate) if cfg.dropout_rate > 0. else nn.Identity()
        self.mlp2_convs = nn.ModuleList()
        self.mlp2_bns = nn.ModuleList()
        if mlp2 is not None:
            for a, b in zip(mlp2[:-1], mlp2[1:]):
                self.mlp2_convs.append(Linear_BN(a, b, bn_ver='1d') if cfg.BATCH_NORM else nn.Linear(a, b))

    def forward(self, sparse_xyz, sparse_feats, nei_inds, sparse_xyz_norm, dense_xyz, dense_xyz_norm, dense_feats=None,
                vi_features=None, inv_neighbors=None, inv_k=None, inv_idx=None):
        nei_inds = nei_inds.contiguous()
        rel, wn_in = _edge_geometry(self.cfg.USE_VI is True, sparse_xyz, sparse_xyz_norm, nei_inds, dense_xyz,
                                    dense_xyz_norm, vi_features)
        feat_pe = None
        if self.cfg.USE_PE:
            if rel is None:
                rel, _ = pcf_fused.edge_geometry(sparse_xyz, None, nei_inds, dense_xyz, None)
            feat_pe = self.pe_convs(rel)
        weights = self.weightnet(wn_in)
        y = self._aggregate_linear(sparse_feats, nei_inds, weights, feat_pe, inv_neighbors, inv_k, inv_idx)
        if dense_feats is not None:
            y = y + dense_feats
        y = self.dropout(y)
        for conv in self.mlp2_convs:
            y = _linear_act(conv, y, pcf_fused.ACT_RELU)
        return y, wn_in
