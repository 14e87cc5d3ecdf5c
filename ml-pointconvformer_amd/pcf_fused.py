"""Differentiable wrappers over the per-edge HIP helpers of libpcf_hip.so (csrc/edge_ops.hip):
row gather, gather+max and the fused edge-geometry / VI kernel.  Used by pcf_layers; there is no
PyTorch fallback (pcf_cuda fails to import without the library)."""
from __future__ import annotations

import ctypes

import torch

import pcf_cuda
from pcf_cuda import _P, _I, _check_input, _floats, _guard, _lib, _ptr, _stream


def _call(fn, *args):
    return pcf_cuda._call(fn, *args)

_LL = ctypes.c_longlong
_Z = ctypes.c_size_t
_F = ctypes.c_float


# BatchNorm `num_batches_tracked` counters: one fused multi-tensor add per flush instead of one tiny kernel per
# BatchNorm call (196 launches per training iteration of the 10cm-lite model).  The layer modules flush at the end
# of their forward; `flush_bn_counters()` is idempotent and cheap when nothing is pending.
_PENDING_COUNTERS = []
_SCOPE_DEPTH = 0


def count_batch(bn):
    if bn.num_batches_tracked is not None:
        _PENDING_COUNTERS.append(bn.num_batches_tracked)
        if _SCOPE_DEPTH == 0 or len(_PENDING_COUNTERS) >= 256:      # functional use: update at once
            flush_bn_counters()


def flush_bn_counters():
    if _PENDING_COUNTERS:
        torch._foreach_add_(_PENDING_COUNTERS, 1)
        _PENDING_COUNTERS.clear()


class CounterScope(torch.nn.Module):
    """Base class of the layer / model modules: the pending BatchNorm counters are flushed when the OUTERMOST such
    module finishes its forward, so a standalone Linear_BN updates its counter at once and a whole model does it in
    one launch."""

    def __call__(self, *args, **kwargs):
        global _SCOPE_DEPTH
        _SCOPE_DEPTH += 1
        try:
            return super().__call__(*args, **kwargs)
        finally:
            _SCOPE_DEPTH -= 1
            if _SCOPE_DEPTH == 0:
                flush_bn_counters()


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
        with _guard(table.device):
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
        with _guard(grad.device):
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
        with _guard(table.device):
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
        with _guard(grad.device):
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
    with _guard(dev):
        _call(_edge_geometry, _ptr(ref_xyz), _ptr(ref_norm) if want_vi else None, _ptr(idx), _ptr(ctr_xyz),
              _ptr(ctr_norm) if want_vi else None, _ptr(rel), _ptr(vi), B, N, M, K, _stream(dev))
    return rel, vi


def vi_from_gathered(localized_xyz, gathered_norm, ctr_norm):
    _floats(localized_xyz=localized_xyz, gathered_norm=gathered_norm, ctr_norm=ctr_norm)
    B, M, K, _ = localized_xyz.shape
    vi = torch.empty(B, M, K, 12, dtype=torch.float32, device=localized_xyz.device)
    with _guard(vi.device):
        _call(_vi_from_gathered, _ptr(localized_xyz), _ptr(gathered_norm), _ptr(ctr_norm), _ptr(vi), B, M, K,
              _stream(vi.device))
    return vi


# --------------------------------------------------------------------------------------------------
# per-edge MLP layer: y = act(BN(x W^T + b))   (csrc/edge_mlp.hip)
# --------------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3
ROWLIN_MAX_CHANNELS = 64

_rowlin_ws = getattr(_lib, 'pcf_hip_rowlin_workspace_bytes')
_rowlin_ws.argtypes = [_I, _I]
_rowlin_ws.restype = _Z
_rowlin_stats = _sig('pcf_hip_rowlin_bn_stats_ex', [_P, _LL, _I, _P, _P, _I, _F, _F, _P, _P, _P, _P,
                                                    _P, _P, _LL, _I, _I, _P, _Z, _P])
_rowlin_fwd = _sig('pcf_hip_rowlin_forward_ex', [_P, _LL, _I, _P, _P, _I, _P, _P, _P, _P, _I,
                                                 _P, _P, _LL, _I, _I, _P, _P])
_rowlin_bwd = _sig('pcf_hip_rowlin_backward_ex', [_P, _P, _LL, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I,
                                                  _P, _P, _LL, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P])
_gdiff_fwd = _sig('pcf_hip_guidance_diff_forward', [_P] * 5 + [_I] * 7 + [_P])
_gdiff_bwd = _sig('pcf_hip_guidance_diff_backward', [_P] * 5 + [_I] * 6 + [_P])


def bn_momentum(bn):
    """The factor of this step's running-statistics update: bn.momentum, or the cumulative average
    1 / (num_batches_tracked + 1) when it is None (torch.nn.modules.batchnorm._BatchNorm.forward).  The counter is
    read on the host only in that (non-default) case."""
    if bn.momentum is not None:
        return float(bn.momentum)
    flush_bn_counters()
    return 1.0 / (float(bn.num_batches_tracked) + 1.0) if bn.num_batches_tracked is not None else 0.0


def same_bn_hyperparameters(bns):
    """The fused chains take ONE eps and ONE momentum for all their BatchNorms: they apply only when the modules agree
    (and use the plain exponential update, momentum not None)."""
    return all(isinstance(b, torch.nn.modules.batchnorm._BatchNorm) and b.momentum is not None and b.eps == bns[0].eps
               and b.momentum == bns[0].momentum and b.track_running_stats == bns[0].track_running_stats
               and not cross_rank_bn(b) for b in bns)


# --------------------------------------------------------------------------------------------------
# SyncBatchNorm (every BASELINE YAML sets sync_bn: True; train_ScanNet_DDP_WarmUP.py:192-193 converts the model)
# --------------------------------------------------------------------------------------------------
_SYNC_WARNED = False


def cross_rank_bn(bn):
    """True when `bn` is a torch.nn.SyncBatchNorm whose batch statistics span more than one rank right now: the fused
    kernels take batch statistics on-rank, so such a module goes through `sync_bn_act` (statistics exchanged with
    torch.distributed) and is kept out of the fused chains.  In eval mode, or with one rank, SyncBatchNorm IS
    BatchNorm and the fused path applies."""
    global _SYNC_WARNED
    if not isinstance(bn, torch.nn.SyncBatchNorm) or not (bn.training or bn.running_mean is None):
        return False
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(bn.process_group) < 2:
        return False
    if not _SYNC_WARNED:
        _SYNC_WARNED = True
        import warnings
        warnings.warn('pcf_layers: SyncBatchNorm over %d ranks -- batch statistics are exchanged per BatchNorm layer '
                      '(as torch.nn.SyncBatchNorm does) and the fused edge chains, whose statistics are rank-local, '
                      'are not used.  Set `sync_bn: False` for the fused path (per-rank statistics over the packed '
                      'batch of each rank).' % dist.get_world_size(bn.process_group))
    return True


class _SyncBN(torch.autograd.Function):
    """y = BN(z) over the last axis with statistics over ALL ranks of `group` (torch.nn.SyncBatchNorm semantics:
    per-rank mean / M2 / count gathered and merged; backward all-reduces sum(dy) and sum(dy * xhat)).  Device-agnostic
    host logic over torch.distributed -- runs on RCCL (GPU) and gloo (CPU tests)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, eps, momentum, group):
        import torch.distributed as dist
        C = z.shape[-1]
        zr = z.reshape(-1, C)
        n = zr.shape[0]
        mean_l = zr.mean(0) if n else zr.new_zeros(C)
        m2_l = ((zr - mean_l) ** 2).sum(0)
        packed = torch.cat([mean_l, m2_l, zr.new_full((1,), float(n))])
        world = dist.get_world_size(group)
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed, group=group)
        allp = torch.stack(parts)
        cnt = allp[:, 2 * C]
        total = cnt.sum()
        mean = (allp[:, :C] * cnt[:, None]).sum(0) / total
        m2 = (allp[:, C:2 * C] + cnt[:, None] * (allp[:, :C] - mean) ** 2).sum(0)
        var = m2 / total
        rstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                running_var.mul_(1 - momentum).add_(m2 / torch.clamp(total - 1, min=1), alpha=momentum)
        xhat = (z - mean) * rstd
        ctx.save_for_backward(xhat, gamma, rstd, total)
        ctx.group = group
        return xhat * gamma + beta

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        xhat, gamma, rstd, total = ctx.saved_tensors
        C = dy.shape[-1]
        dyr, xr = dy.reshape(-1, C), xhat.reshape(-1, C)
        sums = torch.cat([dyr.sum(0), (dyr * xr).sum(0)])
        dbeta, dgamma = sums[:C].clone(), sums[C:].clone()          # rank-local: DDP averages parameter gradients
        dist.all_reduce(sums, group=ctx.group)
        dz = (dy - sums[:C] / total - xhat * (sums[C:] / total)) * (gamma * rstd)
        return dz, dgamma, dbeta, None, None, None, None, None


def sync_bn_act(z, bn, act):
    """act(bn(z)) for a SyncBatchNorm whose statistics span several ranks (see cross_rank_bn)."""
    momentum = bn_momentum(bn)
    if bn.running_mean is not None:
        count_batch(bn)
    y = _SyncBN.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, bn.process_group)
    return _act_torch(y, act)


def _act_torch(y, act):
    if act == ACT_RELU:
        return torch.relu(y)
    if act == ACT_LEAKY:
        return torch.nn.functional.leaky_relu(y, 0.1)
    if act == ACT_SIGMOID:
        return torch.sigmoid(y)
    return y


def rowlin_supported(cin, cout):
    """Layers the fused row-linear kernels are the better choice for: both widths <= 64 and at most four
    16x16 weight tiles (the matrix-core kernels keep W in registers); wider products go through the
    contraction + column-wise BatchNorm kernels."""
    return 1 <= cin <= ROWLIN_MAX_CHANNELS and 1 <= cout <= ROWLIN_MAX_CHANNELS \
        and ((cin + 15) // 16) * ((cout + 15) // 16) <= 4


class _LinearBNAct(torch.autograd.Function):
    """x [..., Cin] -> act(BN(x W^T + b)) [..., Cout]; BN over every axis but the last (batch
    statistics when `training`, running statistics otherwise; gamma None = no BN)."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, running_mean, running_var, eps, momentum, training, act,
                gadd=None, gidx=None, group=0):
        x = x.contiguous()
        W, b = W.contiguous(), b.contiguous()
        Cout, Cin = W.shape
        R = x.numel() // Cin
        dev = x.device
        bn = gamma is not None
        if gadd is not None:
            gadd = gadd.contiguous()
            gN = gadd.shape[-2]
            rpb = gidx[0].numel()
        else:
            gN, rpb = 0, 0
        extras = (_ptr(gadd), _ptr(gidx) if gadd is not None else None, rpb, gN, int(group))
        mean = rstd = None
        stream = _stream(dev)
        with _guard(dev):
            if bn:
                if training:
                    mean = torch.empty(Cout, dtype=torch.float32, device=dev)
                    rstd = torch.empty(Cout, dtype=torch.float32, device=dev)
                    nbytes = _rowlin_ws(Cin, Cout)
                    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    _call(_rowlin_stats, _ptr(x), R, Cin, _ptr(W), _ptr(b), Cout, float(eps), float(momentum),
                          _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(rstd), *extras,
                          ws.data_ptr(), nbytes, stream)
                else:
                    mean = running_mean
                    rstd = torch.rsqrt(running_var + eps)
            y = torch.empty(*x.shape[:-1], Cout, dtype=torch.float32, device=dev)
            _call(_rowlin_fwd, _ptr(x), R, Cin, _ptr(W), _ptr(b), Cout, _ptr(mean), _ptr(rstd),
                  _ptr(gamma) if bn else None, _ptr(beta) if bn else None, int(act), *extras, _ptr(y), stream)
        ctx.save_for_backward(x, W, b, gamma, beta, mean, rstd, gadd, gidx if gadd is not None else None)
        ctx.cfg = (bool(training), int(act), bn, rpb, gN, int(group))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, b, gamma, beta, mean, rstd, gadd, gidx = ctx.saved_tensors
        training, act, bn, rpb, gN, group = ctx.cfg
        dy = dy.contiguous()
        Cout, Cin = W.shape
        R = x.numel() // Cin
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W)
        db = torch.empty_like(b)
        dgamma = torch.empty_like(gamma) if bn else None
        dbeta = torch.empty_like(beta) if bn else None
        dgadd = torch.empty_like(gadd) if gadd is not None else None
        nbytes = _rowlin_ws(Cin, Cout)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _guard(dev):
            _call(_rowlin_bwd, _ptr(x), _ptr(dy), R, Cin, _ptr(W), _ptr(b), Cout, _ptr(mean), _ptr(rstd),
                  _ptr(gamma) if bn else None, _ptr(beta) if bn else None, 1 if training else 0, act,
                  _ptr(gadd), _ptr(gidx), rpb, gN, group,
                  _ptr(dx), _ptr(dW), _ptr(db), _ptr(dgamma), _ptr(dbeta), _ptr(dgadd), ws.data_ptr(), nbytes, _stream(dev))
        return dx, dW, db, dgamma, dbeta, None, None, None, None, None, None, dgadd, None, None


def linear_bn_act(x, weight, bias, bn, act, training, gadd=None, gidx=None, group=0):
    """Fused Linear (+BatchNorm1d module `bn`, or None) (+activation) on the last axis of x.

    Optional extras (first layer of the guidance MLP, Cout <= 16): ``gadd`` [B,N,Cout] is gathered
    through ``gidx`` [B,M,K] and added to x.W^T; ``group`` = K subtracts the pre-bias value of the first
    row of every group of K rows (key = neighbour 0)."""
    x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()   # slices of a parameter are fine
    _floats(x=x, weight=weight, bias=bias)
    if x.shape[-1] != weight.shape[1]:
        raise RuntimeError(f'linear_bn_act: input has {x.shape[-1]} channels, weight expects {weight.shape[1]}')
    if gadd is not None:
        gadd = gadd.contiguous()
        _floats(gadd=gadd)
        _check_input(gidx, 'gidx', torch.int64)
    if bn is None:
        return _LinearBNAct.apply(x, weight, bias, None, None, None, None, 0.0, 0.0, False, act, gadd, gidx, group)
    if cross_rank_bn(bn):
        z = _LinearBNAct.apply(x, weight, bias, None, None, None, None, 0.0, 0.0, False, ACT_NONE, gadd, gidx, group)
        return sync_bn_act(z, bn, act)
    use_batch = training or bn.running_mean is None
    momentum = bn_momentum(bn) if training else 0.0     # before the counter moves (torch: factor = 1 / new count)
    if training:
        count_batch(bn)
    return _LinearBNAct.apply(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum,
                              use_batch, act, gadd, gidx, group)


def split_guidance_supported(K, cout):
    """The gathered-term / key-subtraction form needs K a power of two <= 64 and Cout <= 16."""
    return 1 <= K <= 64 and (K & (K - 1)) == 0 and cout <= 16


class _GuidanceDiff(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gx, idx, pe, use_max):
        B, N, G = gx.shape
        _, M, K, P = pe.shape
        dev = gx.device
        s = torch.empty(B, M, K, G + P, dtype=torch.float32, device=dev)
        argk = torch.empty(B, M, G + P, dtype=torch.uint8, device=dev) if use_max else None
        with _guard(dev):
            _call(_gdiff_fwd, _ptr(gx), _ptr(idx), _ptr(pe), _ptr(s), _ptr(argk), B, N, M, K, G, P, 1 if use_max else 0,
                  _stream(dev))
        ctx.save_for_backward(idx, argk)
        ctx.dims = (B, N, M, K, G, P)
        return s

    @staticmethod
    def backward(ctx, ds):
        idx, argk = ctx.saved_tensors
        B, N, M, K, G, P = ctx.dims
        ds = ds.contiguous()
        dgx = torch.empty(B, N, G, dtype=torch.float32, device=ds.device)
        dpe = torch.empty(B, M, K, P, dtype=torch.float32, device=ds.device)
        with _guard(ds.device):
            _call(_gdiff_bwd, _ptr(ds), _ptr(idx), _ptr(argk), _ptr(dgx), _ptr(dpe), B, N, M, K, G, P, _stream(ds.device))
        return dgx, None, dpe, None


def guidance_diff(guidance_x, nei_inds, feat_pe, use_max):
    """cat(gather(guidance_x), feat_pe) - key, key = neighbour 0 or the max over K (layers.py:372-381)."""
    _floats(guidance_x=guidance_x, feat_pe=feat_pe)
    _check_input(nei_inds, 'nei_inds', torch.int64)
    return _GuidanceDiff.apply(guidance_x, nei_inds, feat_pe, bool(use_max))


# --------------------------------------------------------------------------------------------------
# wide point-level Linear (+BN) (+activation): MFMA contraction + column-wise BN kernels (csrc/bnact.hip)
# --------------------------------------------------------------------------------------------------
_gemm_nt_c = _sig('pcf_hip_gemm_nt', [_P] * 4 + [_I] * 3 + [_P])
_bnact_ws = getattr(_lib, 'pcf_hip_bnact_workspace_bytes')
_bnact_ws.argtypes = [_LL, _I]
_bnact_ws.restype = _Z
_bnact_stats = _sig('pcf_hip_bnact_stats', [_P, _LL, _I, _F, _F, _P, _P, _P, _P, _P, _Z, _P])
_bnact_fwd = _sig('pcf_hip_bnact_forward_res', [_P, _P, _LL, _I, _P, _P, _P, _P, _I, _P, _P])
_bnact_bwd = _sig('pcf_hip_bnact_backward_res', [_P, _P, _P, _LL, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P])
_linbwd_ws = getattr(_lib, 'pcf_hip_linear_backward_workspace_bytes')
_linbwd_ws.argtypes = [_LL, _I, _I]
_linbwd_ws.restype = _Z
_linbwd = _sig('pcf_hip_linear_backward', [_P, _P, _P, _LL, _I, _I, _P, _P, _P, _P, _Z, _P])


class _WideLinearBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, running_mean, running_var, eps, momentum, training, act, residual=None):
        x, W, b = x.contiguous(), W.contiguous(), b.contiguous()
        residual = residual.contiguous() if residual is not None else None
        Cout, Cin = W.shape
        R = x.numel() // Cin
        dev = x.device
        bn = gamma is not None
        stream = _stream(dev)
        z = torch.empty(*x.shape[:-1], Cout, dtype=torch.float32, device=dev)
        mean = rstd = None
        with _guard(dev):
            _call(_gemm_nt_c, _ptr(x), _ptr(W), _ptr(b), _ptr(z), R, Cout, Cin, stream)
            if bn and training:
                mean = torch.empty(Cout, dtype=torch.float32, device=dev)
                rstd = torch.empty(Cout, dtype=torch.float32, device=dev)
                nbytes = _bnact_ws(R, Cout)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                _call(_bnact_stats, _ptr(z), R, Cout, float(eps), float(momentum), _ptr(running_mean), _ptr(running_var),
                      _ptr(mean), _ptr(rstd), ws.data_ptr(), nbytes, stream)
            elif bn:
                mean, rstd = running_mean, torch.rsqrt(running_var + eps)
            if bn or act != ACT_NONE or residual is not None:
                y = torch.empty_like(z)
                _call(_bnact_fwd, _ptr(z), _ptr(residual), R, Cout, _ptr(mean), _ptr(rstd), _ptr(gamma) if bn else None,
                      _ptr(beta) if bn else None, int(act), _ptr(y), stream)
            else:
                y = z
        ctx.save_for_backward(x, W, z, gamma, beta, mean, rstd, residual)
        ctx.cfg = (bool(training), int(act), bn)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, z, gamma, beta, mean, rstd, residual = ctx.saved_tensors
        training, act, bn = ctx.cfg
        dy = dy.contiguous()
        dres = None
        Cout, Cin = W.shape
        R = x.numel() // Cin
        dev = x.device
        stream = _stream(dev)
        dgamma = torch.empty_like(gamma) if bn else None
        dbeta = torch.empty_like(beta) if bn else None
        # a bias in front of a batch-statistics BatchNorm has an identically zero gradient (the mean subtraction cancels
        # it): no column sums of dz; the BN-backward finalize kernel writes the zeros
        zero_db = bn and training
        db = torch.empty(Cout, dtype=torch.float32, device=dev)
        with _guard(dev):
            if bn or act != ACT_NONE:
                dz = torch.empty_like(z)
                if residual is not None and ctx.needs_input_grad[11]:
                    dres = torch.empty_like(z)
                nbytes = _bnact_ws(R, Cout)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                _call(_bnact_bwd, _ptr(z), _ptr(residual), _ptr(dy), R, Cout, _ptr(mean), _ptr(rstd),
                      _ptr(gamma) if bn else None, _ptr(beta) if bn else None, 1 if training else 0, act, _ptr(dz),
                      _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(db) if zero_db else None, ws.data_ptr(), nbytes, stream)
            else:
                dz = dy
                dres = dy if residual is not None else None
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            dW = torch.empty_like(W)
            nbytes = _linbwd_ws(R, Cin, Cout)
            ws2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _call(_linbwd, _ptr(dz), _ptr(x), _ptr(W), R, Cin, Cout, _ptr(dx), _ptr(dW), None if zero_db else _ptr(db),
                  ws2.data_ptr(), nbytes, stream)
        return dx, dW, db, dgamma, dbeta, None, None, None, None, None, None, dres


def wide_linear_bn_act(x, weight, bias, bn, act, training, residual=None):
    """Linear (+BatchNorm1d module `bn` or None) (+residual) (+activation) for channel counts beyond the per-edge
    engine's 64: MFMA contraction + column-wise BN kernels, fully on HIP.  y = act(BN(x W^T + b) + residual)."""
    x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
    _floats(x=x, weight=weight, bias=bias)
    if residual is not None:
        _floats(residual=residual)
        if tuple(residual.shape) != tuple(x.shape[:-1]) + (weight.shape[0],):
            raise RuntimeError('wide_linear_bn_act: residual must have the shape of the output')
    if x.numel() // x.shape[-1] >= 2 ** 31:
        raise RuntimeError('wide_linear_bn_act: more than 2^31 rows')
    if bn is None:
        return _WideLinearBNAct.apply(x, weight, bias, None, None, None, None, 0.0, 0.0, False, act, residual)
    if cross_rank_bn(bn):
        z = _WideLinearBNAct.apply(x, weight, bias, None, None, None, None, 0.0, 0.0, False, ACT_NONE, None)
        y = sync_bn_act(z, bn, ACT_NONE if residual is not None else act)
        return y if residual is None else _act_torch(y + residual, act)
    use_batch = training or bn.running_mean is None
    momentum = bn_momentum(bn) if training else 0.0     # before the counter moves (torch: factor = 1 / new count)
    if training:
        count_batch(bn)
    return _WideLinearBNAct.apply(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum,
                                  use_batch, act, residual)


class _BNAct(torch.autograd.Function):
    """act(BN(z)) over the last axis with the column-wise kernels of csrc/bnact.hip (no contraction in front)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, eps, momentum, training, act):
        z = z.contiguous()
        C = z.shape[-1]
        R = z.numel() // C
        dev = z.device
        stream = _stream(dev)
        y = torch.empty_like(z)
        with _guard(dev):
            if training:
                mean = torch.empty(C, dtype=torch.float32, device=dev)
                rstd = torch.empty(C, dtype=torch.float32, device=dev)
                nbytes = _bnact_ws(R, C)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                _call(_bnact_stats, _ptr(z), R, C, float(eps), float(momentum), _ptr(running_mean), _ptr(running_var),
                      _ptr(mean), _ptr(rstd), ws.data_ptr(), nbytes, stream)
            else:
                mean, rstd = running_mean, torch.rsqrt(running_var + eps)
            _call(_bnact_fwd, _ptr(z), None, R, C, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), int(act), _ptr(y), stream)
        ctx.save_for_backward(z, gamma, beta, mean, rstd)
        ctx.cfg = (bool(training), int(act))
        return y

    @staticmethod
    def backward(ctx, dy):
        z, gamma, beta, mean, rstd = ctx.saved_tensors
        training, act = ctx.cfg
        dy = dy.contiguous()
        C = z.shape[-1]
        R = z.numel() // C
        dev = z.device
        dz = torch.empty_like(z)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        nbytes = _bnact_ws(R, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _guard(dev):
            _call(_bnact_bwd, _ptr(z), None, _ptr(dy), R, C, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
                  1 if training else 0, act, _ptr(dz), None, _ptr(dgamma), _ptr(dbeta), None, ws.data_ptr(), nbytes,
                  _stream(dev))
        return dz, dgamma, dbeta, None, None, None, None, None, None


def bn_act(z, bn, act, training):
    """act(bn(z)) for an nn.BatchNorm1d module over the last axis of z, in three HBM passes forward (statistics,
    normalise + activate) and two backward.  Same statistics rules as nn.BatchNorm1d (layers.py:709, 887, 1082 apply the
    module to the channel axis of the fused aggregate + linear output; the ReLU follows at :721, :901, :1094)."""
    _floats(z=z)
    if bn.weight is None:
        raise RuntimeError('bn_act: BatchNorm without affine parameters is not covered')
    if cross_rank_bn(bn):
        return sync_bn_act(z, bn, act)
    use_batch = training or bn.running_mean is None
    momentum = bn_momentum(bn) if training else 0.0
    if training and bn.running_mean is not None:
        count_batch(bn)
    return _BNAct.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, use_batch, act)


# --------------------------------------------------------------------------------------------------
# fused forward of the PCFLayer edge graph (csrc/edge_chain.hip) + aggregate; backward layer by layer
# --------------------------------------------------------------------------------------------------
_PP = ctypes.POINTER(ctypes.c_void_p)
_chain_ws = getattr(_lib, 'pcf_hip_pcf_chain_workspace_bytes')
_chain_ws.argtypes = []
_chain_ws.restype = _Z
_chain_fwd = _sig('pcf_hip_pcf_chain_forward',
                  [_P, _P, _P, _LL, _LL, _I, _I, _I, _I, _I, _I, _PP, _PP, _PP, _PP, _PP, _PP, _F, _F, _I, _P,
                   _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P])


_chain_bwd_ws = getattr(_lib, 'pcf_hip_pcf_chain_backward_workspace_bytes')
_chain_bwd_ws.argtypes = [_LL]
_chain_bwd_ws.restype = _Z
_chain_fwd_mk = _sig('pcf_hip_pcf_chain_forward_maxkey',
                     [_P, _P, _P, _P, _LL, _LL, _I, _I, _I, _I, _I, _I, _PP, _PP, _PP, _PP, _PP, _PP, _F, _F, _I, _P,
                      _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P])
_chain_bwd_mk = _sig('pcf_hip_pcf_chain_backward_maxkey',
                     [_P, _P, _P, _P, _P, _P, _P, _P, _LL, _LL, _I, _I, _I, _I, _I, _I, _PP, _PP, _PP, _PP, _P, _P, _PP, _PP, _PP,
                      _PP, _P, _Z, _P])
_chain_bwd = _sig('pcf_hip_pcf_chain_backward',
                  [_P, _P, _P, _P, _P, _P, _LL, _LL, _I, _I, _I, _I, _I, _I, _PP, _PP, _PP, _PP, _P, _P, _PP, _PP, _PP, _PP,
                   _P, _Z, _P])


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _rowlin_backward_raw(x, dy, W, b, mean, rstd, gamma, beta, training, act, need_dx, gadd=None, gidx=None, group=0):
    """One layer of the chain backwards through csrc/edge_mlp.hip -> (dx, dW, db, dgamma, dbeta, dgadd)."""
    Cout, Cin = W.shape
    R = x.numel() // Cin
    dev = x.device
    dx = torch.empty_like(x) if need_dx else None
    dW, db = torch.empty_like(W), torch.empty_like(b)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
    dgadd = torch.empty_like(gadd) if gadd is not None else None
    rpb = gidx[0].numel() if gadd is not None else 0
    gN = gadd.shape[-2] if gadd is not None else 0
    nbytes = _rowlin_ws(Cin, Cout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _call(_rowlin_bwd, _ptr(x), _ptr(dy), R, Cin, _ptr(W), _ptr(b), Cout, mean.data_ptr(), rstd.data_ptr(), _ptr(gamma),
          _ptr(beta), 1 if training else 0, act, _ptr(gadd), _ptr(gidx) if gadd is not None else None, rpb, gN, group,
          _ptr(dx), _ptr(dW), _ptr(db), _ptr(dgamma), _ptr(dbeta), _ptr(dgadd), ws.data_ptr(), nbytes, _stream(dev))
    return dx, dW, db, dgamma, dbeta, dgadd


class _PCFChain(torch.autograd.Function):
    """agg = aggregate(fx, idx, score, w) with score / w produced by the fused edge chain.

    Tensor inputs: vi [B,M,K,cv], u [B,N,8], fx [B,N,Ci], then (W, b, gamma, beta) of the six layers in the
    order mlp_conv, g1 (positional half of the weight), g2, w1, w2, w3.  `bns` are the six BatchNorm1d modules
    (running statistics are updated in place in training).  ukey [B,M,8] (or None): strided layers, key = maximum of the
    query over the neighbourhood -- the gathered half of that key, Wa . max_k guidance_x[idx], formed by the caller."""

    @staticmethod
    def forward(ctx, idx, bns, training, fused_backward, ukey, vi, u, fx, *params):
        dev = vi.device
        B, M, K, cv = vi.shape
        N = u.shape[1]
        Ws, bs, gammas, betas = params[0::4], params[1::4], params[2::4], params[3::4]
        g, heads, cm = Ws[0].shape[0], Ws[2].shape[0], Ws[5].shape[0]
        E = B * M * K
        stats = torch.empty(12, 64, dtype=torch.float32, device=dev)
        if not training:
            for l, bn in enumerate(bns):
                c = bn.running_mean.numel()
                stats[l, :c] = bn.running_mean
                stats[6 + l, :c] = torch.rsqrt(bn.running_var + bn.eps)
        f32 = dict(dtype=torch.float32, device=dev)
        score = torch.empty(B, M, K, heads, **f32)
        w = torch.empty(B, M, K, cm, **f32)
        pe = a1 = h1 = a2 = h1_acc = a2_acc = None
        if training:
            if fused_backward:          # the fused backward restarts from the raw accumulators of g1 and w2
                h1_acc, a2_acc = torch.empty(B, M, K, 8, **f32), torch.empty(B, M, K, 8, **f32)
            else:                       # the layer-at-a-time backward reads the intermediate activations
                pe, a1 = torch.empty(B, M, K, g, **f32), torch.empty(B, M, K, 8, **f32)
                h1, a2 = torch.empty(B, M, K, 8, **f32), torch.empty(B, M, K, 8, **f32)
            for bn in bns:
                count_batch(bn)
        nbytes = _chain_ws()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        keep = [t.contiguous() for t in params]
        Ws, bs, gammas, betas = keep[0::4], keep[1::4], keep[2::4], keep[3::4]
        rm = _ptr_array([bn.running_mean for bn in bns]) if training else None
        rv = _ptr_array([bn.running_var for bn in bns]) if training else None
        mom = bns[0].momentum               # same_bn_hyperparameters(): one eps / momentum for the chain, not None
        if ukey is not None:
            ukey = ukey.contiguous()
        with _guard(dev):
            tail = (_ptr(vi), _ptr(idx), _ptr(u), E, M * K, N, K, cv, g, heads, cm, _ptr_array(Ws), _ptr_array(bs),
                    _ptr_array(gammas), _ptr_array(betas), rm, rv, float(bns[0].eps), float(mom), 1 if training else 0,
                    stats.data_ptr(), _ptr(pe), _ptr(a1), _ptr(h1), _ptr(a2), _ptr(h1_acc), _ptr(a2_acc), _ptr(score), _ptr(w),
                    ws.data_ptr(), nbytes, _stream(dev))
            if ukey is None:
                _call(_chain_fwd, *tail)
            else:
                _call(_chain_fwd_mk, _ptr(ukey), *tail)
        agg = pcf_cuda.pcf_forward(fx, idx, score, w)
        ctx.save_for_backward(idx, vi, u, fx, stats, pe, a1, h1, a2, h1_acc, a2_acc, score, w, ukey, *keep)
        ctx.training = bool(training)
        ctx.fused_backward = bool(fused_backward)
        return agg

    @staticmethod
    def backward(ctx, dagg):
        idx, vi, u, fx, stats, pe, a1, h1, a2, h1_acc, a2_acc, score, w, ukey, *keep = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError('pcf_chain: backward needs the training-mode forward (batch statistics)')
        Ws, bs, gammas, betas = keep[0::4], keep[1::4], keep[2::4], keep[3::4]
        tr = ctx.training
        B, M, K, cv = vi.shape
        st = lambda l: (stats[l], stats[6 + l])
        if ctx.fused_backward:
            dev = dagg.device
            with _guard(dev):
                dfx, dscore, dw = pcf_cuda.pcf_backward(dagg.contiguous(), fx, idx, score, w)
                du = torch.empty_like(u)
                grads = [torch.empty_like(t) for t in keep]
                nbytes = _chain_bwd_ws(B * M * K)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                tail = (_ptr(vi), _ptr(idx), _ptr(h1_acc), _ptr(a2_acc), _ptr(dscore), _ptr(dw), B * M * K, M * K,
                        u.shape[1], K, cv,
                        Ws[0].shape[0], Ws[2].shape[0], Ws[5].shape[0], _ptr_array(Ws), _ptr_array(bs), _ptr_array(gammas),
                        _ptr_array(betas), stats.data_ptr(), _ptr(du), _ptr_array(grads[0::4]), _ptr_array(grads[1::4]),
                        _ptr_array(grads[2::4]), _ptr_array(grads[3::4]), ws.data_ptr(), nbytes, _stream(dev))
                dukey = None
                if ukey is None:
                    _call(_chain_bwd, *tail)
                else:
                    dukey = torch.empty_like(ukey)
                    _call(_chain_bwd_mk, _ptr(ukey), _ptr(dukey), *tail)
            return (None, None, None, None, dukey, None, du, dfx, *grads)
        if ukey is not None:
            raise RuntimeError('pcf_chain: the maximum-key form has the fused backward only')
        with _guard(dagg.device):
            dfx, dscore, dw = pcf_cuda.pcf_backward(dagg.contiguous(), fx, idx, score, w)
            L = {}
            dh1, *L[2] = _rowlin_backward_raw(h1, dscore, Ws[2], bs[2], *st(2), gammas[2], betas[2], tr, ACT_SIGMOID, True)[:5]
            r = _rowlin_backward_raw(pe, dh1, Ws[1], bs[1], *st(1), gammas[1], betas[1], tr, ACT_RELU, True, gadd=u, gidx=idx, group=K)
            dpe, L[1], du = r[0], list(r[1:5]), r[5]
            L[0] = list(_rowlin_backward_raw(vi, dpe, Ws[0], bs[0], *st(0), gammas[0], betas[0], tr, ACT_RELU, False)[1:5])
            da2, *L[5] = _rowlin_backward_raw(a2, dw, Ws[5], bs[5], *st(5), gammas[5], betas[5], tr, ACT_RELU, True)[:5]
            da1, *L[4] = _rowlin_backward_raw(a1, da2, Ws[4], bs[4], *st(4), gammas[4], betas[4], tr, ACT_RELU, True)[:5]
            L[3] = list(_rowlin_backward_raw(vi, da1, Ws[3], bs[3], *st(3), gammas[3], betas[3], tr, ACT_RELU, False)[1:5])
        grads = []
        for l in range(6):
            grads.extend(L[l])
        return (None, None, None, None, None, None, du, dfx, *grads)


_wn_fwd = _sig('pcf_hip_weightnet_chain_forward',
               [_P, _LL, _I, _I, _PP, _PP, _PP, _PP, _PP, _PP, _F, _F, _I, _P, _P, _P, _P, _Z, _P])
_wn_bwd = _sig('pcf_hip_weightnet_chain_backward',
               [_P, _P, _P, _LL, _I, _I, _PP, _PP, _PP, _PP, _P, _PP, _PP, _PP, _PP, _P, _Z, _P])


class _WeightNetChain(torch.autograd.Function):
    """w = WeightNet(x) through the fused three-layer chain (csrc/edge_chain*.hip, WeightNet branch only).
    Tensor inputs: x [..., cin], then (W, b, gamma, beta) of w1, w2, w3."""

    @staticmethod
    def forward(ctx, bns, training, x, *params):
        dev = x.device
        cin = x.shape[-1]
        E = x.numel() // cin
        keep = [t.contiguous() for t in params]
        Ws, bs, gammas, betas = keep[0::4], keep[1::4], keep[2::4], keep[3::4]
        cm = Ws[2].shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        stats = torch.empty(12, 64, **f32)
        if not training:
            for i, bn in enumerate(bns):
                c = bn.running_mean.numel()
                stats[3 + i, :c] = bn.running_mean
                stats[9 + i, :c] = torch.rsqrt(bn.running_var + bn.eps)
        w = torch.empty(*x.shape[:-1], cm, **f32)
        a2_acc = torch.empty(*x.shape[:-1], 8, **f32) if training else None
        if training:
            for bn in bns:
                count_batch(bn)
        nbytes = _chain_ws()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rm = _ptr_array([bn.running_mean for bn in bns]) if training else None
        rv = _ptr_array([bn.running_var for bn in bns]) if training else None
        mom = bns[0].momentum               # same_bn_hyperparameters(): one eps / momentum for the chain, not None
        with _guard(dev):
            _call(_wn_fwd, _ptr(x), E, cin, cm, _ptr_array(Ws), _ptr_array(bs), _ptr_array(gammas), _ptr_array(betas), rm, rv,
                  float(bns[0].eps), float(mom), 1 if training else 0, stats.data_ptr(), _ptr(a2_acc), _ptr(w),
                  ws.data_ptr(), nbytes, _stream(dev))
        ctx.save_for_backward(x, stats, a2_acc, *keep)
        ctx.training = bool(training)
        return w

    @staticmethod
    def backward(ctx, dw):
        x, stats, a2_acc, *keep = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError('weightnet_chain: backward needs the training-mode forward (batch statistics)')
        Ws, bs, gammas, betas = keep[0::4], keep[1::4], keep[2::4], keep[3::4]
        dev = dw.device
        cin = x.shape[-1]
        E = x.numel() // cin
        dw = dw.contiguous()
        grads = [torch.empty_like(t) for t in keep]
        nbytes = _chain_bwd_ws(E)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _guard(dev):
            _call(_wn_bwd, _ptr(x), _ptr(a2_acc), _ptr(dw), E, cin, Ws[2].shape[0], _ptr_array(Ws), _ptr_array(bs),
                  _ptr_array(gammas), _ptr_array(betas), stats.data_ptr(), _ptr_array(grads[0::4]), _ptr_array(grads[1::4]),
                  _ptr_array(grads[2::4]), _ptr_array(grads[3::4]), ws.data_ptr(), nbytes, _stream(dev))
        return (None, None, None, *grads)


def weightnet_chain_supported(cin, hidden, cout, n_rows):
    """Three Linear_BN layers cin -> 8 -> 8 -> cout with cin <= 12, cout <= 16, a multiple of 16 rows."""
    return tuple(hidden) == (8, 8) and 1 <= cin <= 12 and 1 <= cout <= 16 and n_rows % 16 == 0 and n_rows > 0


def weightnet_chain(x, layers, training):
    """layers: three (nn.Linear, nn.BatchNorm1d) pairs.  The input carries no gradient (coordinates / VI)."""
    _floats(x=x)
    params = []
    for lin, bn in layers:
        params += [lin.weight, lin.bias, bn.weight, bn.bias]
    return _WeightNetChain.apply([bn for _, bn in layers], training, x.contiguous(), *params)


class _SplitColumns(torch.autograd.Function):
    """W [O, A+B] -> (W[:, :A], W[:, A:]) as contiguous tensors; the backward is one concatenation instead of two
    zero-fill + copy + accumulate chains of the slicing ops."""

    @staticmethod
    def forward(ctx, W, A):
        return W[:, :A].contiguous(), W[:, A:].contiguous()

    @staticmethod
    def backward(ctx, dA, dB):
        return torch.cat([dA, dB], dim=1), None


def split_columns(W, A):
    return _SplitColumns.apply(W, A)


def pcf_chain_supported(cv, g, heads, cm, K, hidden_ok, n_edges, edges_per_batch=16):
    """Shapes the fused MFMA chain covers: a neighbourhood must fit one 16-edge tile."""
    return hidden_ok and 1 <= cv <= 12 and 1 <= g <= 32 and 1 <= heads <= 8 and 1 <= cm <= 16 \
        and 1 <= K <= 16 and (K & (K - 1)) == 0 and n_edges % 16 == 0 and edges_per_batch >= 16


def pcf_chain(vi, idx, u, fx, layers, training, fused_backward=True, g1_positional_weight=None, ukey=None):
    """layers: six (nn.Linear, nn.BatchNorm1d) pairs in the order mlp_conv, g1, g2, w1, w2, w3; the g1 weight is
    split here (its gathered half already went into `u`).  fused_backward: adjoint through the three-pass
    recompute kernel (csrc/edge_chain_bwd.hip); False keeps the activations and goes layer by layer."""
    _floats(vi=vi, u=u, fx=fx)
    _check_input(idx, 'nei_inds', torch.int64)
    G = layers[0][0].out_features
    params = []
    for l, (lin, bn) in enumerate(layers):
        W = lin.weight
        if l == 1:      # positional half of the first guidance layer (the gathered half already went into `u`)
            W = g1_positional_weight if g1_positional_weight is not None else lin.weight[:, lin.weight.shape[1] - G:]
        params += [W, lin.bias, bn.weight, bn.bias]
    return _PCFChain.apply(idx, [bn for _, bn in layers], training, fused_backward, ukey, vi, u, fx, *params)


# --------------------------------------------------------------------------------------------------
# point-level Linear + BatchNorm chains of a PCFLayer (csrc/fused_linear.hip): the "head" in front of the edge graph
# (unary1 -> guidance_unary -> per-point half of the first guidance layer) and the "tail" behind the aggregate
# (linear + ReLU -> unary2 -> + shortcut -> LeakyReLU), training mode.  17 launches per step instead of ~45.
# --------------------------------------------------------------------------------------------------
_flin_ws = getattr(_lib, 'pcf_hip_flin_workspace_bytes')
_flin_ws.argtypes = [_LL, _I, _I]
_flin_ws.restype = _Z
_flin_ticket_ints = getattr(_lib, 'pcf_hip_flin_ticket_ints')
_flin_ticket_ints.argtypes = []
_flin_ticket_ints.restype = _I
_flin_fwd = _sig('pcf_hip_flin_forward', [_P, _LL, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _Z, _P, _P])
_flin_bwd_in = _sig('pcf_hip_flin_backward_input', [_P, _P, _P, _I, _LL, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _P, _P])
_flin_bwd_w = _sig('pcf_hip_flin_backward_weight_slabs', [_P, _P, _P, _I, _P, _P, _I, _LL, _I, _I, _P, _Z, _P])
_flin_bwd_w_splits = _sig('pcf_hip_flin_backward_weight_splits', [_LL, _I, _I])
_slab_sum_multi = _sig('pcf_hip_slab_sum_multi', [_I, _PP, _PP, ctypes.POINTER(_LL), ctypes.POINTER(_I), _P])
_bn_bwd_stats = _sig('pcf_hip_bn_backward_stats', [_P, _P, _P, _P, _I, _LL, _I, _P, _P, _P, _P, _P, _Z, _P, _P])

_TICKETS = {}
_IDENTITY_CST = {}


def _tickets(dev):
    """The persistent, zero-initialised ticket buffer of `dev` (the kernels leave it zeroed)."""
    t = _TICKETS.get(dev)
    if t is None:
        t = _TICKETS[dev] = torch.zeros(_flin_ticket_ints(), dtype=torch.int32, device=dev)
    return t


def _identity_cst(dev, C):
    """Record of a layer without BatchNorm and activation: sc = 1, sh = 0, mean = 0, rstd = 1, D1 = D0 = 0."""
    t = _IDENTITY_CST.get((dev, C))
    if t is None:
        t = torch.zeros(6, C, dtype=torch.float32, device=dev)
        t[0] = 1.0
        t[3] = 1.0
        _IDENTITY_CST[(dev, C)] = t
    return t


def _ws(dev, rows, cout, cin):
    nbytes = _flin_ws(rows, cout, cin)
    return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes


def _flin_forward(x2d, pre, pre_act, side, W, b, bn, momentum, stream, dev):
    """z, cst of one layer (batch statistics, running statistics updated)."""
    R, K = x2d.shape
    N = W.shape[0]
    z = torch.empty(R, N, dtype=torch.float32, device=dev)
    cst = torch.empty(6, N, dtype=torch.float32, device=dev) if bn is not None else None
    ws, nbytes = _ws(dev, R, N, K)
    _call(_flin_fwd, _ptr(x2d), R, K, _ptr(pre), int(pre_act), _ptr(side), _ptr(W), _ptr(b), N, _ptr(z), _ptr(cst),
          _ptr(bn.weight) if bn is not None else None, _ptr(bn.bias) if bn is not None else None,
          _ptr(bn.running_mean) if bn is not None else None, _ptr(bn.running_var) if bn is not None else None,
          float(bn.eps) if bn is not None else 0.0, float(momentum), ws.data_ptr(), nbytes, _tickets(dev).data_ptr(), stream)
    return z, cst


def _flin_bwd_input(dy, z, cst, act, W, add, zp, cstp, actp, grads_p, stream, dev):
    """dx = dz W (+ add); with a producer (zp, cstp, actp) its dgamma / dbeta / zero bias gradient go to grads_p."""
    R, K = dy.shape
    N = W.shape[1]
    dx = torch.empty(R, N, dtype=torch.float32, device=dev)
    ws, nbytes = _ws(dev, R, K, N)
    dg, dbe, dbi = grads_p if grads_p is not None else (None, None, None)
    _call(_flin_bwd_in, _ptr(dy), _ptr(z), _ptr(cst), int(act), R, K, _ptr(W), N, _ptr(add), _ptr(dx), _ptr(zp), _ptr(cstp),
          int(actp), _ptr(dg), _ptr(dbe), _ptr(dbi), ws.data_ptr(), nbytes, _tickets(dev).data_ptr(), stream)
    return dx


class _WeightGrads:
    """Weight gradients of a chain: each product leaves its row-range slabs, one launch sums them all at the end."""

    def __init__(self, dev, stream):
        self.dev, self.stream, self.items = dev, stream, []

    def add(self, dy, z, cst, act, xin, pre, pre_act):
        R, M = dy.shape
        N = xin.shape[1]
        splits = _flin_bwd_w_splits(R, M, N)
        slabs = torch.empty(splits, M, N, dtype=torch.float32, device=self.dev)
        dW = torch.empty(M, N, dtype=torch.float32, device=self.dev)
        _call(_flin_bwd_w, _ptr(dy), _ptr(z), _ptr(cst), int(act), _ptr(xin), _ptr(pre), int(pre_act), R, M, N, slabs.data_ptr(),
              slabs.numel() * 4, self.stream)
        self.items.append((slabs, dW, M * N, splits))
        return dW

    def finish(self):
        n = len(self.items)
        _call(_slab_sum_multi, n, _ptr_array([t[0] for t in self.items]), _ptr_array([t[1] for t in self.items]),
              (_LL * n)(*[t[2] for t in self.items]), (_I * n)(*[t[3] for t in self.items]), self.stream)
        self.items = []


class _PointHead(torch.autograd.Function):
    """(fx, u) = head(x): fx = LeakyReLU(BN1(x W1^T + b1)) (or x when the block has no unary1), guidance_x = BN2(fx W2^T + b2),
    u = guidance_x Wa^T.  Three launches forward, six backward; keeps z1, z2 (raw) and two small records."""

    @staticmethod
    def forward(ctx, bns, x, Wa, W2, b2, g2, be2, W1=None, b1=None, g1=None, be1=None):
        dev = x.device
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        bn1, bn2 = bns
        stream = _stream(dev)
        with _guard(dev):
            if W1 is not None:
                z1, cst1 = _flin_forward(x2, None, 0, None, W1, b1, bn1, bn_momentum(bn1), stream, dev)
                count_batch(bn1)
                fx = torch.empty(x2.shape[0], W1.shape[0], dtype=torch.float32, device=dev)
                z2, cst2 = _flin_forward(z1, cst1, ACT_LEAKY, fx, W2, b2, bn2, bn_momentum(bn2), stream, dev)
            else:
                z1 = cst1 = None
                fx = x2
                z2, cst2 = _flin_forward(x2, None, 0, None, W2, b2, bn2, bn_momentum(bn2), stream, dev)
            count_batch(bn2)
            u, _ = _flin_forward(z2, cst2, ACT_NONE, None, Wa, None, None, 0.0, stream, dev)
        ctx.save_for_backward(x2, z1, cst1, z2, cst2, u, Wa, W2, W1)
        ctx.shape = shape
        return fx.view(*shape[:-1], fx.shape[-1]), u.view(*shape[:-1], u.shape[-1])

    @staticmethod
    def backward(ctx, dfx, du):
        x2, z1, cst1, z2, cst2, u, Wa, W2, W1 = ctx.saved_tensors
        dev = x2.device
        stream = _stream(dev)
        R = x2.shape[0]
        du2 = du.reshape(R, -1).contiguous()
        dfx2 = dfx.reshape(R, -1).contiguous() if dfx is not None else None
        f32 = dict(dtype=torch.float32, device=dev)
        ident = _identity_cst(dev, Wa.shape[0])
        G = W2.shape[0]
        dg2, dbe2, db2 = torch.empty(G, **f32), torch.empty(G, **f32), torch.empty(G, **f32)
        with _guard(dev):
            wg = _WeightGrads(dev, stream)
            dWa = wg.add(du2, du2, ident, ACT_NONE, z2, cst2, ACT_NONE)
            dgx = _flin_bwd_input(du2, du2, ident, ACT_NONE, Wa, None, z2, cst2, ACT_NONE, (dg2, dbe2, db2), stream, dev)
            if W1 is not None:
                mid = W1.shape[0]
                dg1, dbe1, db1 = torch.empty(mid, **f32), torch.empty(mid, **f32), torch.empty(mid, **f32)
                dW2 = wg.add(dgx, z2, cst2, ACT_NONE, z1, cst1, ACT_LEAKY)
                dfx_t = _flin_bwd_input(dgx, z2, cst2, ACT_NONE, W2, dfx2, z1, cst1, ACT_LEAKY, (dg1, dbe1, db1), stream, dev)
                dW1 = wg.add(dfx_t, z1, cst1, ACT_LEAKY, x2, None, 0)
                dx = _flin_bwd_input(dfx_t, z1, cst1, ACT_LEAKY, W1, None, None, None, 0, None, stream, dev) \
                    if ctx.needs_input_grad[1] else None
                tail = (dW1, db1, dg1, dbe1)
            else:
                dW2 = wg.add(dgx, z2, cst2, ACT_NONE, x2, None, 0)
                dx = _flin_bwd_input(dgx, z2, cst2, ACT_NONE, W2, dfx2, None, None, 0, None, stream, dev) \
                    if ctx.needs_input_grad[1] else None
                tail = ()
            wg.finish()
        if dx is not None:
            dx = dx.view(ctx.shape)
        return (None, dx, dWa, dW2, db2, dg2, dbe2, *tail)


class _PointTail(torch.autograd.Function):
    """out = LeakyReLU(BN4(ReLU(BN3(agg W3^T + b3)) W4^T + b4) + shortcut).  Three launches forward, five backward."""

    @staticmethod
    def forward(ctx, bns, agg, shortcut, W3, b3, g3, be3, W4, b4, g4, be4):
        dev = agg.device
        shape = agg.shape
        a2 = agg.reshape(-1, shape[-1])
        bn3, bn4 = bns
        stream = _stream(dev)
        sc2 = shortcut.reshape(a2.shape[0], -1)
        with _guard(dev):
            z3, cst3 = _flin_forward(a2, None, 0, None, W3, b3, bn3, bn_momentum(bn3), stream, dev)
            count_batch(bn3)
            z4, cst4 = _flin_forward(z3, cst3, ACT_RELU, None, W4, b4, bn4, bn_momentum(bn4), stream, dev)
            count_batch(bn4)
            out = torch.empty_like(z4)
            R, C = z4.shape
            _call(_bnact_fwd, _ptr(z4), _ptr(sc2), R, C, cst4[2].data_ptr(), cst4[3].data_ptr(), _ptr(g4), _ptr(be4),
                  ACT_LEAKY, _ptr(out), stream)
        ctx.save_for_backward(a2, sc2, z3, cst3, z4, cst4, W3, W4)
        ctx.shape = shape
        return out.view(*shape[:-1], out.shape[-1])

    @staticmethod
    def backward(ctx, dout):
        a2, sc2, z3, cst3, z4, cst4, W3, W4 = ctx.saved_tensors
        dev = a2.device
        stream = _stream(dev)
        R = a2.shape[0]
        d2 = dout.reshape(R, -1).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        C4, C3 = W4.shape[0], W3.shape[0]
        dg4, dbe4, db4 = torch.empty(C4, **f32), torch.empty(C4, **f32), torch.empty(C4, **f32)
        dg3, dbe3, db3 = torch.empty(C3, **f32), torch.empty(C3, **f32), torch.empty(C3, **f32)
        g = torch.empty_like(z4)
        with _guard(dev):
            ws, nbytes = _ws(dev, R, C4, C4)
            _call(_bn_bwd_stats, _ptr(d2), _ptr(z4), _ptr(sc2), _ptr(cst4), ACT_LEAKY, R, C4, _ptr(g), _ptr(dg4), _ptr(dbe4),
                  _ptr(db4), ws.data_ptr(), nbytes, _tickets(dev).data_ptr(), stream)
            wg = _WeightGrads(dev, stream)
            dW4 = wg.add(g, z4, cst4, ACT_NONE, z3, cst3, ACT_RELU)
            dy3 = _flin_bwd_input(g, z4, cst4, ACT_NONE, W4, None, z3, cst3, ACT_RELU, (dg3, dbe3, db3), stream, dev)
            dW3 = wg.add(dy3, z3, cst3, ACT_RELU, a2, None, 0)
            dagg = _flin_bwd_input(dy3, z3, cst3, ACT_RELU, W3, None, None, None, 0, None, stream, dev) \
                if ctx.needs_input_grad[1] else None
            wg.finish()
        if dagg is not None:
            dagg = dagg.view(ctx.shape)
        dsc = g.view(*ctx.shape[:-1], C4) if ctx.needs_input_grad[2] else None
        return (None, dagg, dsc, dW3, db3, dg3, dbe3, dW4, db4, dg4, dbe4)


# The contraction kernels with fused statistics (csrc/fused_linear.hip) carry ~10 us of ticketed hand-over per launch: at
# parity with the layer-by-layer kernels on 80k rows (435 vs 450 us per PCFLayer step), slower on the smaller tensors of the
# models' levels -- whole training iterations with them everywhere / from 32768 rows / never: 10cm-lite 21.58 / 21.39 / 21.15 ms,
# PCF_Normal 26.99 / 26.66 / 26.32 ms, 5cm 26.65 / 26.55 / 26.34 ms.  So below this row count only the row chains of
# csrc/point_chain.hip (whole layer chains per pass, at the widths they are instantiated for) replace the layer-by-layer path.
FLIN_CHAIN_MIN_ROWS = 65536


def head_chain_pays(x, unary1, guidance_unary, wa_rows=8, force=False):
    rows = x.numel() // max(1, x.shape[-1])
    if force:
        return True
    if unary1 is not None and wa_rows == 8 and rows > 0 and \
            point_head_chain_supported(unary1.mlp.c.in_features, unary1.mlp.c.out_features, guidance_unary.mlp.c.out_features):
        return True
    return rows >= FLIN_CHAIN_MIN_ROWS


def tail_chain_pays(agg, linear, unary2, force=False):
    rows = agg.numel() // max(1, agg.shape[-1])
    if force:
        return True
    if rows > 0 and point_tail_chain_supported(linear.c.in_features, linear.c.out_features, unary2.mlp.c.out_features):
        return True
    return rows >= FLIN_CHAIN_MIN_ROWS


def point_chain_ok(*bns):
    """The fused point-level chains apply in training mode to plain (rank-local) BatchNorms with the exponential
    running-statistics update."""
    return all(isinstance(b, torch.nn.modules.batchnorm._BatchNorm) and b.training and b.momentum is not None
               and b.track_running_stats and b.weight is not None and not cross_rank_bn(b) for b in bns)


def point_head(x, unary1, guidance_unary, Wa):
    """(fx, u) for a PCFLayer: unary1 (UnaryBlock with BatchNorm, or None when the block has none), guidance_unary
    (UnaryBlock, no activation), Wa = the gathered half of the first guidance layer's weight."""
    _floats(x=x)
    l2 = guidance_unary.mlp
    args = [Wa.contiguous(), l2.c.weight, l2.c.bias, l2.bn.weight, l2.bn.bias]
    bn1 = None
    if unary1 is not None:
        l1 = unary1.mlp
        bn1 = l1.bn
        args += [l1.c.weight, l1.c.bias, l1.bn.weight, l1.bn.bias]
        if x.numel() > 0 and bn1.eps == l2.bn.eps and Wa.shape[0] == 8 and \
                point_head_chain_supported(l1.c.in_features, l1.c.out_features, l2.c.out_features):
            return _PointHeadChain.apply((bn1, l2.bn), x.contiguous(), *args)      # narrow widths: the row chain
    return _PointHead.apply((bn1, l2.bn), x.contiguous(), *args)


def point_tail(agg, shortcut, linear, unary2):
    """LeakyReLU(unary2(ReLU(linear(agg))) + shortcut) for two Linear_BN modules (layers.py:393-414)."""
    _floats(agg=agg, shortcut=shortcut)
    l4 = unary2.mlp
    fn = _PointTail
    if agg.numel() > 0 and linear.bn.eps == l4.bn.eps and point_tail_chain_supported(linear.c.in_features, linear.c.out_features, l4.c.out_features):
        fn = _PointTailChain                                                     # BASELINE widths: the row chain
    return fn.apply((linear.bn, l4.bn), agg.contiguous(), shortcut.contiguous(), linear.c.weight, linear.c.bias,
                            linear.bn.weight, linear.bn.bias, l4.c.weight, l4.c.bias, l4.bn.weight, l4.bn.bias)


# --------------------------------------------------------------------------------------------------
# attention arithmetic of the ablation layers (csrc/attention_ops.hip)
# --------------------------------------------------------------------------------------------------
_sm_agg_fwd = _sig('pcf_hip_softmax_aggregate_forward', [_P, _P, _P, _P, _LL, _I, _I, _I, _P])
_sm_agg_bwd = _sig('pcf_hip_softmax_aggregate_backward', [_P, _P, _P, _P, _P, _LL, _I, _I, _I, _P])
_qk_fwd = _sig('pcf_hip_qk_score_forward', [_P, _P, _P, _LL, _I, _I, _I, _F, _P])
_qk_bwd = _sig('pcf_hip_qk_score_backward', [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _F, _P])
_ln_fwd = _sig('pcf_hip_layer_norm_forward', [_P, _P, _P, _P, _P, _P, _LL, _I, _F, _P])
_ln_bwd_ws = getattr(_lib, 'pcf_hip_layer_norm_backward_workspace_bytes')
_ln_bwd_ws.argtypes = [_LL, _I]
_ln_bwd_ws.restype = _Z
_ln_bwd = _sig('pcf_hip_layer_norm_backward', [_P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _P, _Z, _P])


class _SoftmaxAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, logit):
        v, logit = v.contiguous(), logit.contiguous()
        B, M, K, C = v.shape
        J = logit.shape[-1]
        dev = v.device
        out = torch.empty(B, M, C, dtype=torch.float32, device=dev)
        sm = torch.empty_like(logit)
        with _guard(dev):
            _call(_sm_agg_fwd, _ptr(v), _ptr(logit), _ptr(out), _ptr(sm), B * M, K, C, J, _stream(dev))
        ctx.save_for_backward(v, sm)
        return out

    @staticmethod
    def backward(ctx, dout):
        v, sm = ctx.saved_tensors
        B, M, K, C = v.shape
        J = sm.shape[-1]
        dev = v.device
        dv, dl = torch.empty_like(v), torch.empty_like(sm)
        with _guard(dev):
            _call(_sm_agg_bwd, _ptr(dout.contiguous()), _ptr(v), _ptr(sm), _ptr(dv), _ptr(dl), B * M, K, C, J, _stream(dev))
        return dv, dl


def softmax_aggregate(v, logit):
    """out[b,m,c] = sum_k v[b,m,k,c] * softmax_k(logit)[b,m,k,c % J]   (PointTransformerLayer, layers.py:519-527)."""
    _floats(v=v.contiguous(), logit=logit.contiguous())
    if v.dim() != 4 or logit.dim() != 4 or v.shape[:3] != logit.shape[:3] or v.shape[3] % logit.shape[3] != 0:
        raise RuntimeError('softmax_aggregate: expected v [B,M,K,C] and logit [B,M,K,J] with J dividing C')
    return _SoftmaxAggregate.apply(v, logit)


class _QKScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, key, scale):
        q, key = q.contiguous(), key.contiguous()
        B, N, K, H, D = q.shape
        dev = q.device
        score = torch.empty(B, N, K, H, dtype=torch.float32, device=dev)
        with _guard(dev):
            _call(_qk_fwd, _ptr(q), _ptr(key), _ptr(score), B * N, K, H, D, float(scale), _stream(dev))
        ctx.save_for_backward(q, key, score)
        ctx.scale = float(scale)
        return score

    @staticmethod
    def backward(ctx, ds):
        q, key, score = ctx.saved_tensors
        B, N, K, H, D = q.shape
        dev = q.device
        dq, dkey = torch.empty_like(q), torch.empty_like(key)
        with _guard(dev):
            _call(_qk_bwd, _ptr(ds.contiguous()), _ptr(score), _ptr(q), _ptr(key), _ptr(dq), _ptr(dkey), B * N, K, H, D, ctx.scale,
                  _stream(dev))
        return dq, dkey, None


def qk_score(q, key, scale):
    """sigmoid(scale * <q[b,n,k,h,:], key[b,n,h,:]>) -> [B,N,K,H]   (MultiHeadGuidanceQK, layers.py:100-114)."""
    _floats(q=q.contiguous(), key=key.contiguous())
    return _QKScore.apply(q, key, scale)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        C = x.shape[-1]
        R = x.numel() // C
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(R, dtype=torch.float32, device=dev)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        with _guard(dev):
            _call(_ln_fwd, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), R, C, float(eps), _stream(dev))
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        C = x.shape[-1]
        R = x.numel() // C
        dev = x.device
        dx = torch.empty_like(x)
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        nbytes = _ln_bwd_ws(R, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _guard(dev):
            _call(_ln_bwd, _ptr(dy.contiguous()), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dg), _ptr(db), R, C,
                  ws.data_ptr(), nbytes, _stream(dev))
        return dx, dg, db, None


def layer_norm(x, ln):
    """ln(x) for an nn.LayerNorm over the last axis (elementwise_affine), on HIP."""
    _floats(x=x.contiguous())
    if len(ln.normalized_shape) != 1 or ln.normalized_shape[0] != x.shape[-1] or ln.weight is None or ln.bias is None:
        raise RuntimeError('layer_norm: LayerNorm over the last axis with affine parameters expected')
    return _LayerNorm.apply(x, ln.weight, ln.bias, ln.eps)


# --------------------------------------------------------------------------------------------------
# PCFLayer head as a row chain on the matrix cores (csrc/point_chain.hip), for the widths it is instantiated for
# --------------------------------------------------------------------------------------------------
_ph_ok = _sig('pcf_hip_point_head_supported', [_I, _I, _I])
_ph_ws = getattr(_lib, 'pcf_hip_point_head_workspace_bytes')
_ph_ws.argtypes = [_LL, _I, _I, _I]
_ph_ws.restype = _Z
_ph_fwd = _sig('pcf_hip_point_head_forward', [_P, _LL, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _F, _P, _F,
                                              _P, _P, _P, _P, _P, _P, _Z, _P, _P])
_ph_bwd = _sig('pcf_hip_point_head_backward', [_P, _P, _P, _P, _P, _LL, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                               _P, _P, _P, _P, _Z, _P, _P])


def point_head_chain_supported(c_in, mid, g):
    return bool(_ph_ok(int(c_in), int(mid), int(g)))


class _PointHeadChain(torch.autograd.Function):
    """(fx, u) = head(x) through the three-pass row chain (csrc/point_chain.hip): three launches forward, four backward."""

    @staticmethod
    def forward(ctx, bns, x, Wa, W2, b2, g2, be2, W1, b1, g1, be1):
        dev = x.device
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        R, cin = x2.shape
        mid, G = W1.shape[0], W2.shape[0]
        bn1, bn2 = bns
        f32 = dict(dtype=torch.float32, device=dev)
        z1, fx = torch.empty(R, mid, **f32), torch.empty(R, mid, **f32)
        u = torch.empty(R, Wa.shape[0], **f32)
        cst1, cst2 = torch.empty(6, mid, **f32), torch.empty(6, G, **f32)
        nbytes = _ph_ws(R, cin, mid, G)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        m1, m2 = bn_momentum(bn1), bn_momentum(bn2)
        with _guard(dev):
            _call(_ph_fwd, _ptr(x2), R, cin, mid, G, _ptr(W1), _ptr(b1), _ptr(g1), _ptr(be1), _ptr(bn1.running_mean),
                  _ptr(bn1.running_var), float(m1), _ptr(W2), _ptr(b2), _ptr(g2), _ptr(be2), _ptr(bn2.running_mean),
                  _ptr(bn2.running_var), float(m2), _ptr(Wa), float(bn1.eps), _ptr(z1), _ptr(fx), _ptr(u), _ptr(cst1), _ptr(cst2),
                  ws.data_ptr(), nbytes, _tickets(dev).data_ptr(), _stream(dev))
        count_batch(bn1)
        count_batch(bn2)
        ctx.save_for_backward(x2, z1, fx, cst1, cst2, Wa, W2, b2, W1)
        ctx.shape = shape
        return fx.view(*shape[:-1], mid), u.view(*shape[:-1], u.shape[-1])

    @staticmethod
    def backward(ctx, dfx, du):
        x2, z1, fx, cst1, cst2, Wa, W2, b2, W1 = ctx.saved_tensors
        dev = x2.device
        R, cin = x2.shape
        mid, G = W1.shape[0], W2.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        du2 = du.reshape(R, -1).contiguous()
        dfx2 = dfx.reshape(R, -1).contiguous() if dfx is not None else None
        dx = torch.empty(R, cin, **f32)
        dW1, dW2, dWa = torch.empty_like(W1), torch.empty_like(W2), torch.empty_like(Wa)
        db1, dg1, dbe1 = torch.empty(mid, **f32), torch.empty(mid, **f32), torch.empty(mid, **f32)
        db2, dg2, dbe2 = torch.empty(G, **f32), torch.empty(G, **f32), torch.empty(G, **f32)
        nbytes = _ph_ws(R, cin, mid, G)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _guard(dev):
            _call(_ph_bwd, _ptr(dfx2), _ptr(du2), _ptr(x2), _ptr(z1), _ptr(fx), R, cin, mid, G, _ptr(W1), _ptr(W2), _ptr(b2), _ptr(Wa),
                  _ptr(cst1), _ptr(cst2), _ptr(dx), _ptr(dW1), _ptr(db1), _ptr(dg1), _ptr(dbe1), _ptr(dW2), _ptr(db2), _ptr(dg2),
                  _ptr(dbe2), _ptr(dWa), ws.data_ptr(), nbytes, _tickets(dev).data_ptr(), _stream(dev))
        return (None, dx.view(ctx.shape), dWa, dW2, db2, dg2, dbe2, dW1, db1, dg1, dbe1)


_pt_ok = _sig('pcf_hip_point_tail_supported', [_I, _I, _I])
_pt_ws = getattr(_lib, 'pcf_hip_point_tail_workspace_bytes')
_pt_ws.argtypes = [_LL, _I, _I, _I]
_pt_ws.restype = _Z
_pt_fwd = _sig('pcf_hip_point_tail_forward', [_P, _LL, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _F, _F,
                                              _P, _P, _P, _P, _P, _Z, _P, _P])
_pt_bwd = _sig('pcf_hip_point_tail_backward', [_P, _P, _P, _P, _LL, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                               _P, _Z, _P, _P])


def point_tail_chain_supported(c_agg, c_half, c_out):
    return bool(_pt_ok(int(c_agg), int(c_half), int(c_out)))


class _PointTailChain(torch.autograd.Function):
    """_PointTail through the row chain (csrc/point_chain.hip): three launches forward, six backward (dW3, the wide weight
    gradient, stays with the contraction kernel of fused_linear.hip)."""

    @staticmethod
    def forward(ctx, bns, agg, shortcut, W3, b3, g3, be3, W4, b4, g4, be4):
        dev = agg.device
        shape = agg.shape
        a2 = agg.reshape(-1, shape[-1])
        R, ca = a2.shape
        ch, co = W3.shape[0], W4.shape[0]
        bn3, bn4 = bns
        sc2 = shortcut.reshape(R, -1)
        f32 = dict(dtype=torch.float32, device=dev)
        z3, z4, out = torch.empty(R, ch, **f32), torch.empty(R, co, **f32), torch.empty(R, co, **f32)
        cst3, cst4 = torch.empty(6, ch, **f32), torch.empty(6, co, **f32)
        nbytes = _pt_ws(R, ca, ch, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        m3, m4 = bn_momentum(bn3), bn_momentum(bn4)
        stream = _stream(dev)
        with _guard(dev):
            _call(_pt_fwd, _ptr(a2), R, ca, ch, co, _ptr(W3), _ptr(b3), _ptr(g3), _ptr(be3), _ptr(bn3.running_mean),
                  _ptr(bn3.running_var), float(m3), _ptr(W4), _ptr(b4), _ptr(g4), _ptr(be4), _ptr(bn4.running_mean),
                  _ptr(bn4.running_var), float(m4), float(bn3.eps), _ptr(z3), _ptr(z4), _ptr(cst3), _ptr(cst4), ws.data_ptr(),
                  nbytes, _tickets(dev).data_ptr(), stream)
            _call(_bnact_fwd, _ptr(z4), _ptr(sc2), R, co, cst4[2].data_ptr(), cst4[3].data_ptr(), _ptr(g4), _ptr(be4),
                  ACT_LEAKY, _ptr(out), stream)
        count_batch(bn3)
        count_batch(bn4)
        ctx.save_for_backward(a2, sc2, z3, cst3, z4, cst4, W3, W4)
        ctx.shape = shape
        return out.view(*shape[:-1], co)

    @staticmethod
    def backward(ctx, dout):
        a2, sc2, z3, cst3, z4, cst4, W3, W4 = ctx.saved_tensors
        dev = a2.device
        stream = _stream(dev)
        R, ca = a2.shape
        ch, co = W3.shape[0], W4.shape[0]
        d2 = dout.reshape(R, -1).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        dg4, dbe4, db4 = torch.empty(co, **f32), torch.empty(co, **f32), torch.empty(co, **f32)
        dg3, dbe3, db3 = torch.empty(ch, **f32), torch.empty(ch, **f32), torch.empty(ch, **f32)
        g4, g3 = torch.empty(R, co, **f32), torch.empty(R, ch, **f32)
        dagg, dW4 = torch.empty(R, ca, **f32), torch.empty_like(W4)
        nbytes = _pt_ws(R, ca, ch, co)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _guard(dev):
            _call(_pt_bwd, _ptr(d2), _ptr(sc2), _ptr(z3), _ptr(z4), R, ca, ch, co, _ptr(W3), _ptr(W4), _ptr(cst3), _ptr(cst4),
                  _ptr(g4), _ptr(g3), _ptr(dagg), _ptr(dW4), _ptr(db3), _ptr(dg3), _ptr(dbe3), _ptr(db4), _ptr(dg4), _ptr(dbe4),
                  ws.data_ptr(), nbytes, _tickets(dev).data_ptr(), stream)
            wg = _WeightGrads(dev, stream)
            dW3 = wg.add(g3, z3, cst3, ACT_RELU, a2, None, 0)      # g3 is masked already; the ReLU mask is idempotent
            wg.finish()
        dsc = g4.view(*ctx.shape[:-1], co) if ctx.needs_input_grad[2] else None
        return (None, dagg.view(ctx.shape), dsc, dW3, db3, dg3, dbe3, dW4, db4, dg4, dbe4)


# --------------------------------------------------------------------------------------------------
# pe_convs of the strided / transposed PointConvs as a row chain over the edges (csrc/point_chain.hip)
# --------------------------------------------------------------------------------------------------
_pe_ok = _sig('pcf_hip_pe_chain_supported', [_I, _I])
_pe_ws = getattr(_lib, 'pcf_hip_pe_chain_workspace_bytes')
_pe_ws.argtypes = [_LL, _I, _I]
_pe_ws.restype = _Z
_pe_fwd = _sig('pcf_hip_pe_chain_forward', [_P, _LL, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P,
                                            _P, _Z, _P, _P])
_pe_bwd = _sig('pcf_hip_pe_chain_backward', [_P, _P, _LL, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _P])


def pe_chain_supported(cin, hidden, cout):
    """Two Linear_BN + ReLU layers 3 -> hidden -> cout at the widths the chain is instantiated for."""
    return cin == 3 and len(hidden) == 1 and bool(_pe_ok(int(hidden[0]), int(cout)))


class _PEChain(torch.autograd.Function):
    """feat_pe = ReLU(BN2(W2 ReLU(BN1(W1 rel + b1)) + b2)), training mode: three launches forward, four backward; only the
    offsets and the output exist in HBM."""

    @staticmethod
    def forward(ctx, bns, rel, W1, b1, g1, be1, W2, b2, g2, be2):
        dev = rel.device
        shape = rel.shape
        r2 = rel.reshape(-1, 3)
        E = r2.shape[0]
        H, L = W1.shape[0], W2.shape[0]
        bn1, bn2 = bns
        f32 = dict(dtype=torch.float32, device=dev)
        out = torch.empty(E, L, **f32)
        cst1, cst2 = torch.empty(6, H, **f32), torch.empty(6, L, **f32)
        nbytes = _pe_ws(E, H, L)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        m1, m2 = bn_momentum(bn1), bn_momentum(bn2)
        with _guard(dev):
            _call(_pe_fwd, _ptr(r2), E, H, L, _ptr(W1), _ptr(b1), _ptr(g1), _ptr(be1), _ptr(bn1.running_mean), _ptr(bn1.running_var),
                  float(m1), _ptr(W2), _ptr(b2), _ptr(g2), _ptr(be2), _ptr(bn2.running_mean), _ptr(bn2.running_var), float(m2),
                  float(bn1.eps), _ptr(out), _ptr(cst1), _ptr(cst2), ws.data_ptr(), nbytes, _tickets(dev).data_ptr(), _stream(dev))
        count_batch(bn1)
        count_batch(bn2)
        ctx.save_for_backward(r2, cst1, cst2, W1, b1, W2, b2)
        return out.view(*shape[:-1], L)

    @staticmethod
    def backward(ctx, dout):
        r2, cst1, cst2, W1, b1, W2, b2 = ctx.saved_tensors
        dev = r2.device
        E = r2.shape[0]
        H, L = W1.shape[0], W2.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        d2 = dout.reshape(E, L).contiguous()
        dW1, dW2 = torch.empty_like(W1), torch.empty_like(W2)
        db1, dg1, dbe1 = torch.empty(H, **f32), torch.empty(H, **f32), torch.empty(H, **f32)
        db2, dg2, dbe2 = torch.empty(L, **f32), torch.empty(L, **f32), torch.empty(L, **f32)
        nbytes = _pe_ws(E, H, L)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _guard(dev):
            _call(_pe_bwd, _ptr(d2), _ptr(r2), E, H, L, _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(cst1), _ptr(cst2), _ptr(dW1),
                  _ptr(db1), _ptr(dg1), _ptr(dbe1), _ptr(dW2), _ptr(db2), _ptr(dg2), _ptr(dbe2), ws.data_ptr(), nbytes,
                  _tickets(dev).data_ptr(), _stream(dev))
        return (None, None, dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2)


def pe_chain(rel, layers):
    """layers: two (nn.Linear, nn.BatchNorm1d) pairs in training mode; the offsets carry no gradient."""
    _floats(rel=rel)
    (l1, bn1), (l2, bn2) = layers
    return _PEChain.apply((bn1, bn2), rel.contiguous(), l1.weight.contiguous(), l1.bias, bn1.weight, bn1.bias, l2.weight.contiguous(),
                          l2.bias, bn2.weight, bn2.bias)


# --------------------------------------------------------------------------------------------------
# cross entropy of the segmentation head (csrc/loss.hip)
# --------------------------------------------------------------------------------------------------
_ce_ws = getattr(_lib, 'pcf_hip_cross_entropy_workspace_bytes')
_ce_ws.argtypes = [_LL]
_ce_ws.restype = _Z
_ce_fwd = _sig('pcf_hip_cross_entropy_forward', [_P, _P, _LL, _I, _LL, _F, _P, _P, _P, _Z, _P])
_ce_bwd = _sig('pcf_hip_cross_entropy_backward', [_P, _P, _P, _LL, _I, _P, _P])


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, label_smoothing):
        dev = logits.device
        R, C = logits.shape
        stat = torch.empty(2, dtype=torch.float32, device=dev)
        dlogits = torch.empty_like(logits)
        nbytes = _ce_ws(R)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _guard(dev):
            _call(_ce_fwd, _ptr(logits), _ptr(target), R, C, int(ignore_index), float(label_smoothing), stat.data_ptr(),
                  _ptr(dlogits), ws.data_ptr(), nbytes, _stream(dev))
        ctx.save_for_backward(dlogits, stat)
        return stat[0]

    @staticmethod
    def backward(ctx, grad_out):
        dlogits, stat = ctx.saved_tensors
        dev = dlogits.device
        R, C = dlogits.shape
        dx = torch.empty_like(dlogits)
        g = grad_out.reshape(1).contiguous().float()
        with _guard(dev):
            _call(_ce_bwd, _ptr(dlogits), _ptr(g), stat.data_ptr(), R, C, _ptr(dx), _stream(dev))
        return dx, None, None, None


def cross_entropy_supported(criterion, logits):
    """nn.CrossEntropyLoss(weight=None, reduction='mean') on float32 HIP logits [R, C <= 64]."""
    return type(criterion) is torch.nn.CrossEntropyLoss and criterion.weight is None and criterion.reduction == 'mean' \
        and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and 1 <= logits.shape[1] <= 64


def cross_entropy(logits, target, ignore_index=-100, label_smoothing=0.0):
    """torch.nn.functional.cross_entropy(logits, target, ignore_index=, label_smoothing=) with mean reduction, in two launches
    forward and one backward (torch: log_softmax + a one-workgroup nll reduction, ~0.28 ms on 144k x 20).
    A target outside [0, C) that is not `ignore_index` is treated like an ignored row (no contribution to loss, count or
    gradient); torch raises a device-side assertion for it -- labels are expected to be validated by the data pipeline."""
    _floats(logits=logits)
    _check_input(target, 'target', torch.int64)
    if target.numel() != logits.shape[0]:
        raise RuntimeError('cross_entropy: one target per row of the logits')
    return _CrossEntropy.apply(logits.contiguous(), target.contiguous(), ignore_index, label_smoothing)
