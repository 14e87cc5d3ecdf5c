"""Alias of the reference's module name ``layer_utils`` (layer_utils.py:13-319) onto ``pcf_layers``: the operator
boundary (autograd Functions + their nn.Module wrappers), ``index_points``, ``VI_coordinate_transform``, ``Linear_BN``
and ``UnaryBlock`` on the HIP kernels.  Nothing is defined in this file."""
from pcf_layers import (PCF, Linear_BN, PCFFunction, PConv, PConvFunction, PConvLinearOpt, PConvLinearOptFunction,  # noqa: F401
                        UnaryBlock, VI_coordinate_transform, index_points)
