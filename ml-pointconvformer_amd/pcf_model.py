"""Backbone + segmentation decoder on the MI355X layers: host-side mirror of the reference's
``model_architecture.py`` (``get_default_configs`` :13-77, ``PCF_Backbone`` :80-245,
``PointConvFormer_Segmentation`` :345-502).

Same constructor arguments, ``forward`` signatures (including the ``inv_*`` CSR triples that
``train_ScanNet_DDP_WarmUP.py:401-404`` passes) and sub-module names, so a reference ``state_dict``
loads with ``strict=True``.  The graph is the reference's:

  level 0          PointConv(6 -> base) + 2 x PointConvStridePE                    (use_level_1)
  level i -> i+1   strided PCFLayer (PointConvStridePE for i <= guided_level) on edges_forward[i]
                   then resblocks[i+1] x same-resolution layers on edges_self[i+1]; the VI features
                   of a resolution are computed by its first block and reused by the others
  decoder          PointConvTransposePE on edges_propagate[level] with the encoder skip
  head             Linear_BN + ReLU + Linear

``cfg.transformer_type`` other than 'PCF' builds the PointTransformerLayer ablation blocks in the guided levels
(model_architecture.py:138-176, 214-237).  ``PCONV_OPT`` and ``USE_CUDA_KERNEL`` are given defaults here (the reference leaves
``PCONV_OPT`` undefaulted and crashes on configs that omit it).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

import pcf_fused
from pcf_layers import (Linear_BN, PCFLayer, PointConv, PointConvStridePE, PointConvTransposePE, PointTransformerLayer,
                        _linear_act)


class Config(dict):
    """Attribute-style dict that raises AttributeError on a missing key, like easydict.EasyDict."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


_DEFAULTS = dict(USE_VI=True, USE_PE=False, transformer_type='PCF', attention_type='subtraction',
                 layer_norm_guidance=False, drop_path_rate=0., BATCH_NORM=True, dropout_rate=0., TIME=False,
                 USE_XYZ=True, point_dim=3, mid_dim_back=1, use_level_1=True, USE_CUDA_KERNEL=True, PCONV_OPT=True,
                 dropout_fc=0.)


def get_default_configs(cfg, num_level=5, base_dim=64):
    """Fill the model defaults in place and return cfg.  (model_architecture.py:13-77)"""
    cfg['num_level'] = num_level
    cfg['base_dim'] = base_dim
    if 'feat_dim' not in cfg:
        cfg['feat_dim'] = [base_dim * (i + 1) for i in range(num_level + 1)]
    for k, v in _DEFAULTS.items():
        if k not in cfg:
            cfg[k] = v
    return cfg


def _inv(triple, i):
    """(inv_neighbors, inv_k, inv_idx) of level i from three per-level lists, or Nones."""
    if triple is None or triple[0] is None:
        return {}
    return dict(inv_neighbors=triple[0][i], inv_k=triple[1][i], inv_idx=triple[2][i])


class PCF_Backbone(pcf_fused.CounterScope):
    def __init__(self, cfg, input_feat_dim=3):
        super().__init__()
        self.cfg = cfg
        self.point_transformer = cfg.transformer_type != 'PCF'

        self.total_level = cfg.num_level
        self.guided_level = cfg.guided_level
        self.input_feat_dim = input_feat_dim + 3 if cfg.USE_XYZ else input_feat_dim
        wn_in = cfg.point_dim + 9 if cfg.USE_VI is True else cfg.point_dim
        if cfg.use_level_1:
            start = [wn_in, cfg.mid_dim[0]]
            self.selfpointconv = PointConv(self.input_feat_dim, cfg.base_dim, cfg, start)
            self.selfpointconv_res1 = PointConvStridePE(cfg.base_dim, cfg.base_dim, cfg, start)
            self.selfpointconv_res2 = PointConvStridePE(cfg.base_dim, cfg.base_dim, cfg, start)
        else:
            self.selfmlp = Linear_BN(self.input_feat_dim, cfg.base_dim, bn_ver='1d')
        self.pointconv = nn.ModuleList()
        self.pointconv_res = nn.ModuleList()
        for i in range(1, self.total_level):
            in_ch, out_ch = cfg.feat_dim[i - 1], cfg.feat_dim[i]
            wn = [wn_in, cfg.mid_dim[i]]
            guided = i > self.guided_level

            def make(a, b):
                if not guided:
                    return PointConvStridePE(a, b, cfg, wn)
                return PointTransformerLayer(a, b, cfg.num_heads) if self.point_transformer else PCFLayer(a, b, cfg, wn, cfg.num_heads)

            self.pointconv.append(make(in_ch, out_ch))
            self.pointconv_res.append(nn.ModuleList(make(out_ch, out_ch) for _ in range(cfg.resblocks[i])))

    def forward(self, features, pointclouds, edges_self, edges_forward, norms, inv_neighbors_self=None,
                inv_k_self=None, inv_idx_self=None, inv_neighbors_forward=None, inv_k_forward=None,
                inv_idx_forward=None):
        inv_self = (inv_neighbors_self, inv_k_self, inv_idx_self)
        inv_fwd = (inv_neighbors_forward, inv_k_forward, inv_idx_forward)
        x = torch.cat([features, pointclouds[0]], -1) if self.cfg.USE_XYZ else features
        if self.cfg.use_level_1:
            a0 = _inv(inv_self, 0) if self.cfg.PCONV_OPT else {}
            x, vi = self.selfpointconv(pointclouds[0], x, edges_self[0], norms[0], **a0)
            x, _ = self.selfpointconv_res1(pointclouds[0], x, edges_self[0], norms[0], vi_features=vi, **a0)
            x, _ = self.selfpointconv_res2(pointclouds[0], x, edges_self[0], norms[0], vi_features=vi, **a0)
        else:
            x = _linear_act(self.selfmlp, x, pcf_fused.ACT_RELU)
        feats = [x]
        for i, down in enumerate(self.pointconv):
            af = _inv(inv_fwd, i) if self.cfg.PCONV_OPT else {}
            if isinstance(down, PointTransformerLayer):       # model_architecture.py:214-216
                x = down(pointclouds[i], feats[-1], edges_forward[i], pointclouds[i + 1])
            else:
                x, _ = down(pointclouds[i], feats[-1], edges_forward[i], norms[i], pointclouds[i + 1], norms[i + 1], **af)
            vi = None          # neighbourhoods change with the resolution: recomputed by the first block
            a_self = _inv(inv_self, i + 1) if self.cfg.PCONV_OPT else {}
            for block in self.pointconv_res[i]:
                if isinstance(block, PointTransformerLayer):  # :225-227
                    x = block(pointclouds[i + 1], x, edges_self[i + 1])
                    continue
                x, vi_new = block(pointclouds[i + 1], x, edges_self[i + 1], norms[i + 1], vi_features=vi, **a_self)
                vi = vi_new if vi is None else vi
            feats.append(x)
        return feats


# Backbone presets (model_architecture.py:248-342): name -> (levels, heads, blocks per level, C_mid, grid-size ratios)
_PRESETS = {
    'PCF_Tiny': (5, 1, [0, 1, 1, 1, 1], 4, [1, 2, 4, 8, 16]),
    'PCF_Small': (5, 8, [0, 2, 2, 2, 2], 4, [1, 2, 4, 8, 16]),
    'PCF_Normal': (5, 8, [0, 2, 4, 6, 6], 16, [1, 2, 4, 8, 16]),
    'PCF_Large': (6, 8, [0, 2, 4, 6, 6, 2], 16, [1, 2.5, 5, 10, 20, 40]),
}


def _preset(name):
    levels, heads, blocks, cmid, ratios = _PRESETS[name]

    def factory(input_grid_size, base_dim=64):
        """-> (PCF_Backbone, cfg) for the voxel size of the finest resolution (0.02 / 0.05 / 0.1 ...)."""
        cfg = get_default_configs(Config(), num_level=levels, base_dim=base_dim)
        cfg.guided_level, cfg.num_heads = 0, heads
        cfg.resblocks = list(blocks)
        cfg.mid_dim = [cmid] * levels
        cfg.grid_size = [input_grid_size * r for r in ratios]
        return PCF_Backbone(cfg), cfg

    factory.__name__ = factory.__qualname__ = name
    return factory


PCF_Tiny, PCF_Small, PCF_Normal, PCF_Large = (_preset(n) for n in ('PCF_Tiny', 'PCF_Small', 'PCF_Normal', 'PCF_Large'))


class PointConvFormer_Segmentation(pcf_fused.CounterScope):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.total_level = cfg.num_level
        self.pcf_backbone = PCF_Backbone(cfg)
        wn = [cfg.point_dim + 9 if cfg.USE_VI is True else cfg.point_dim, cfg.mid_dim_back]
        self.pointdeconv = nn.ModuleList()
        self.pointdeconv_res = nn.ModuleList()
        for i in range(self.total_level - 2, -1, -1):
            in_ch = cfg.feat_dim[i + 1]
            out_ch = cfg.base_dim if i == 0 else cfg.feat_dim[i]
            self.pointdeconv.append(PointConvTransposePE(in_ch, out_ch, cfg, wn, [out_ch, out_ch]))
            # the reference tests resblocks[i] but counts with resblocks_back[i] (model_architecture.py:388-395)
            n_res = 0 if cfg.resblocks[i] == 0 else cfg.resblocks_back[i]
            self.pointdeconv_res.append(nn.ModuleList(PointConvStridePE(out_ch, out_ch, cfg, wn) for _ in range(n_res)))
        self.fc1 = Linear_BN(cfg.base_dim, cfg.base_dim, bn_ver='1d')
        self.dropout_fc = nn.Dropout(p=cfg.dropout_fc) if cfg.dropout_fc > 0. else nn.Identity()
        self.fc2 = nn.Linear(cfg.base_dim, cfg.num_classes)

    def forward(self, features, pointclouds, edges_self, edges_forward, edges_propagate, norms, inv_self=None,
                inv_forward=None, inv_propagate=None):
        opt = bool(self.cfg.PCONV_OPT) and inv_self is not None
        inv_self = inv_self if opt else (None, None, None)
        inv_forward = inv_forward if opt else (None, None, None)
        inv_propagate = inv_propagate if opt else (None, None, None)
        feats = self.pcf_backbone(features, pointclouds, edges_self, edges_forward, norms, *inv_self, *inv_forward)
        x = feats[-1]
        for i, up in enumerate(self.pointdeconv):
            lvl = self.total_level - 2 - i
            x, _ = up(pointclouds[lvl + 1], x, edges_propagate[lvl], norms[lvl + 1], pointclouds[lvl], norms[lvl],
                      feats[lvl], **_inv(inv_propagate, lvl))
            vi = None
            for block in self.pointdeconv_res[i]:
                x, vi_new = block(pointclouds[lvl], x, edges_self[lvl], norms[lvl], vi_features=vi, **_inv(inv_self, lvl))
                vi = vi_new if vi is None else vi
            feats[lvl] = x
        x = self.dropout_fc(_linear_act(self.fc1, x, pcf_fused.ACT_RELU))
        return _linear_act(self.fc2, x, pcf_fused.ACT_NONE)       # the classifier on the same contraction kernels
