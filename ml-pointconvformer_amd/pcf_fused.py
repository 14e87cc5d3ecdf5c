"""Differentiable wrappers over the per-edge HIP helpers of libpcf_hip.so (csrc/edge_ops.hip):
row gather, gather+max and the fused edge-geometry / VI kernel.  Used by pcf_layers; there is no
PyTorch fallback (pcf_cuda fails to import without the library)."""
from __future__ import annotations

import ctypes

import torch

import pcf_cuda
from pcf_cuda import _P, _I, _check_input, _floats, _lib, _ptr, _stream


def _call(fn, *args):
    return pcf_cuda._call(fn, *args)

_LL = ctypes.c_longlong


def _sig(name, argtypes):
    fn = getattr(_lib, name)
    fn.argtypes = argtypes
    fn.restype = _I
    fn.__name__ = name
    return fn


_gather_rows = _sig('pcf_hip_gather_rows', [_P, _P, _P, _I, _I, _LL, _I, _P])
_scatter_add_rows = _sig('pcf_hip_scatter_add_rows', [_P, _P, _P, _I, _I, _LL, _I, _P])
_gather_max = _sig('pcf_hip_gather_max', [_P] * 4 + [_I] * 5 + [_P])
_gather_max_bwd = _sig('pcf_hip_gather_max_backward', [_P] * 4 + [_I] * 5 + [_P])
_edge_geometry = _sig('pcf_hip_edge_geometry', [_P] * 7 + [_I] * 4 + [_P])
_vi_from_gathered = _sig('pcf_hip_vi_from_gathered', [_P] * 4 + [_I] * 3 + [_P])


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, idx):
        B, N, C = table.shape
        S = idx[0].numel()
        out = torch.empty(*idx.shape, C, dtype=table.dtype, device=table.device)
        with torch.cuda.device(table.device):
            _call(_gather_rows, _ptr(table), _ptr(idx), _ptr(out), B, N, S, C, _stream(table.device))
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, S, C)
        return out

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        B, N, S, C = ctx.dims
        grad = grad.contiguous()
        gt = torch.empty(B, N, C, dtype=grad.dtype, device=grad.device)
        with torch.cuda.device(grad.device):
            _call(_scatter_add_rows, _ptr(grad), _ptr(idx), _ptr(gt), B, N, S, C, _stream(grad.device))
        return gt, None


def gather_rows(table, idx):
    """table [B,N,C] f32, idx [B,S] or [B,M,K] i64 -> [B,S,C] / [B,M,K,C]; differentiable in table."""
    _floats(table=table)
    _check_input(idx, 'idx', torch.int64)
    if table.dim() != 3 or idx.dim() not in (2, 3) or idx.shape[0] != table.shape[0]:
        raise RuntimeError('gather_rows: expected table [B,N,C] and idx [B,S] or [B,M,K]')
    return _GatherRows.apply(table, idx)


class _GatherMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, idx):
        B, N, C = table.shape
        _, M, K = idx.shape
        out = torch.empty(B, M, C, dtype=table.dtype, device=table.device)
        argk = torch.empty(B, M, C, dtype=torch.uint8, device=table.device)
        with torch.cuda.device(table.device):
            _call(_gather_max, _ptr(table), _ptr(idx), _ptr(out), _ptr(argk), B, N, M, K, C, _stream(table.device))
        ctx.save_for_backward(idx, argk)
        ctx.dims = (B, N, M, K, C)
        return out

    @staticmethod
    def backward(ctx, grad):
        idx, argk = ctx.saved_tensors
        B, N, M, K, C = ctx.dims
        grad = grad.contiguous()
        gt = torch.empty(B, N, C, dtype=grad.dtype, device=grad.device)
        with torch.cuda.device(grad.device):
            _call(_gather_max_bwd, _ptr(grad), _ptr(idx), _ptr(argk), _ptr(gt), B, N, M, K, C, _stream(grad.device))
        return gt, None


def gather_max(table, idx):
    """max over the K gathered rows: table [B,N,C], idx [B,M,K] -> [B,M,C]  (layers.py:403-408)."""
    _floats(table=table)
    _check_input(idx, 'idx', torch.int64)
    if table.dim() != 3 or idx.dim() != 3 or idx.shape[0] != table.shape[0]:
        raise RuntimeError('gather_max: expected table [B,N,C] and idx [B,M,K]')
    return _GatherMax.apply(table, idx)


def edge_geometry(ref_xyz, ref_norm, idx, ctr_xyz, ctr_norm, want_rel=True):
    """-> (rel [B,M,K,3] or None, vi [B,M,K,12] or None).  Coordinates carry no gradient in the
    reference's models (inputs of the network), so neither output is differentiable."""
    _floats(ref_xyz=ref_xyz, ctr_xyz=ctr_xyz)
    _check_input(idx, 'nei_inds', torch.int64)
    want_vi = ref_norm is not None and ctr_norm is not None
    if want_vi:
        _floats(ref_norm=ref_norm, ctr_norm=ctr_norm)
    B, N, D = ref_xyz.shape
    _, M, K = idx.shape
    if D != 3 or tuple(ctr_xyz.shape) != (B, M, 3):
        raise RuntimeError('edge_geometry: coordinates must be [B,N,3] / [B,M,3] (VI is defined for 3-D only)')
    dev = ref_xyz.device
    rel = torch.empty(B, M, K, 3, dtype=torch.float32, device=dev) if (want_rel or not want_vi) else None
    vi = torch.empty(B, M, K, 12, dtype=torch.float32, device=dev) if want_vi else None
    with torch.cuda.device(dev):
        _call(_edge_geometry, _ptr(ref_xyz), _ptr(ref_norm) if want_vi else None, _ptr(idx), _ptr(ctr_xyz),
              _ptr(ctr_norm) if want_vi else None, _ptr(rel), _ptr(vi), B, N, M, K, _stream(dev))
    return rel, vi


def vi_from_gathered(localized_xyz, gathered_norm, ctr_norm):
    _floats(localized_xyz=localized_xyz, gathered_norm=gathered_norm, ctr_norm=ctr_norm)
    B, M, K, _ = localized_xyz.shape
    vi = torch.empty(B, M, K, 12, dtype=torch.float32, device=localized_xyz.device)
    with torch.cuda.device(vi.device):
        _call(_vi_from_gathered, _ptr(localized_xyz), _ptr(gathered_norm), _ptr(ctr_norm), _ptr(vi), B, M, K,
              _stream(vi.device))
    return vi
