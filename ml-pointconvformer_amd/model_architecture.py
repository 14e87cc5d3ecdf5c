"""Alias of the reference's module name ``model_architecture`` (model_architecture.py:13-502) onto ``pcf_model``:
``from model_architecture import PointConvFormer_Segmentation, get_default_configs`` (train_ScanNet_DDP_WarmUP.py:29-30)
gets the backbone + decoder built on the fused HIP layers.  Nothing is defined in this file."""
from pcf_model import (PCF_Backbone, PCF_Large, PCF_Normal, PCF_Small, PCF_Tiny, PointConvFormer_Segmentation,  # noqa: F401
                       get_default_configs)
from pcf_layers import (Linear_BN, PCFLayer, PointConv, PointConvStridePE, PointConvTransposePE,  # noqa: F401
                        PointTransformerLayer)
