"""Alias of the reference's module name ``layers`` onto this build's layer classes.

With ``ml-pointconvformer_amd/`` in front of the reference checkout on ``PYTHONPATH`` an unchanged
``train_ScanNet_DDP_WarmUP.py`` / ``model_architecture`` import resolves ``layers`` here and gets the fused
HIP-backed classes of ``pcf_layers`` under the reference's names (layers.py:23-1105).  Nothing is defined in this file.
"""
from pcf_layers import (DropPath, MultiHeadGuidance, MultiHeadGuidanceQK, PCFLayer, PointConv, PointConvStridePE,  # noqa: F401
                        PointConvTransposePE, PointTransformerLayer, WeightNet)
from pcf_layers import (PCF, Linear_BN, PConv, PConvLinearOpt, UnaryBlock, VI_coordinate_transform,  # noqa: F401
                        index_points)
