"""``pcf_cuda`` for AMD Instinct MI355X: the reference's operator module, re-implemented on HIP.

Drop-in for the torch extension the reference builds from ``cpp_wrappers/cpp_pcf_kernel``
(``pcf_cuda.cpp:9-19``): same module name, the same nine functions with the same argument
order, dtypes, shapes and return arity, so ``layer_utils.py`` / ``util/common_util.py`` /
``train_ScanNet_DDP_WarmUP.py`` import and call it unchanged.  Put this package's parent
directory (``ml-pointconvformer_amd/``) on ``PYTHONPATH``.

Underneath is ``libpcf_hip.so`` (C ABI in ``include/pcf_hip.h``, hand-written gfx950 kernels in
``csrc/``), bound here with ctypes: tensors go down as raw device pointers, work is enqueued
on torch's *current* HIP stream of the tensors' device (the reference uses the legacy default
stream and no device guard), outputs are allocated with torch.  There is no CPU
or PyTorch fallback: if the library is missing the import fails, and every call raises on
non-device tensors exactly as the reference's CHECK_INPUT does (``pcf.h:14-24``).

Differences from the reference, all deliberate:
  * backward functions return the true adjoint of the forward (forward channel layout
    ``c*C_mid + m``); the reference's CUDA backward indexes ``m*C_in + c``.
  * ``compute_knn_inverse`` orders each CSR bucket by (query, k); the reference's order is
    whatever its atomics produced.
  * ``pconv_linear_opt_backward`` uses the CSR for every input row (deterministic grad_input)
    and performs no device->host sync (the reference syncs twice per call, pconv_ops.cu:887-889).
  * fp32 only (the reference's CUTLASS path is fp32-only too; its older kernels also took fp64).
"""
from __future__ import annotations

import ctypes
import os

import torch

__all__ = [
    'pcf_forward', 'pcf_backward', 'pconv_forward', 'pconv_backward', 'pconv_linear_forward',
    'pconv_linear_backward', 'pconv_linear_opt_backward', 'compute_knn_inverse',
    'pconv_linear_cutlass_forward', 'pcf_backward_csr', 'knn_packed', 'gemm_nt', 'voxelize', 'grid_subsample', 'library_path', 'version',
]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get('PCF_HIP_LIBRARY', os.path.join(_HERE, 'libpcf_hip.so'))

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f'pcf_cuda: {_LIB_PATH} not found. Build it with `make -C ml-pointconvformer_amd/csrc` '
        '(or __graft_entry__.build()); there is no fallback implementation.')
_lib = ctypes.CDLL(_LIB_PATH)

_P, _I, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


def _sig(name, argtypes, restype=_I):
    fn = getattr(_lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
    fn.__name__ = name
    return fn


_version = _sig('pcf_hip_version', [], ctypes.c_char_p)
_last_error = _sig('pcf_hip_last_error', [], ctypes.c_char_p)
_pcf_fwd = _sig('pcf_hip_pcf_forward', [_P] * 5 + [_I] * 7 + [_P])
_pcf_bwd = _sig('pcf_hip_pcf_backward', [_P] * 8 + [_I] * 7 + [_P])
_pcf_bwd_csr_ws = _sig('pcf_hip_pcf_backward_csr_workspace_bytes', [_I] * 4, _Z)
_pcf_bwd_csr = _sig('pcf_hip_pcf_backward_csr', [_P] * 12 + [_Z] + [_I] * 9 + [_P])
_pconv_fwd = _sig('pcf_hip_pconv_forward', [_P] * 5 + [_I] * 7 + [_P])
_pconv_bwd = _sig('pcf_hip_pconv_backward', [_P] * 8 + [_I] * 7 + [_P])
_pl_fwd = _sig('pcf_hip_pconv_linear_forward', [_P] * 8 + [_I] * 8 + [_P])
_pl_bwd_ws = _sig('pcf_hip_pconv_linear_backward_workspace_bytes', [_I] * 8, _Z)
_pl_bwd = _sig('pcf_hip_pconv_linear_backward', [_P] * 13 + [_Z] + [_I] * 8 + [_P])
_plo_bwd_ws = _sig('pcf_hip_pconv_linear_opt_backward_workspace_bytes', [_I] * 8, _Z)
_plo_bwd = _sig('pcf_hip_pconv_linear_opt_backward', [_P] * 16 + [_Z] + [_I] * 10 + [_P])
_inv_ws = _sig('pcf_hip_knn_inverse_workspace_bytes', [_I] * 4, _Z)
_inv = _sig('pcf_hip_knn_inverse', [_P] * 5 + [_Z] + [_I] * 4 + [_P])
_knn = _sig('pcf_hip_knn', [_P] * 4 + [_I] * 3 + [_P] * 2)
_knn_grid_ws = _sig('pcf_hip_knn_grid_workspace_bytes', [_I, _I], _Z)
_IP = ctypes.POINTER(ctypes.c_int)
_PP = ctypes.POINTER(ctypes.c_void_p)
_inv_batch_ws = _sig('pcf_hip_knn_inverse_batched_workspace_bytes', [_I, _IP, _IP, _IP], _Z)
_inv_batch = _sig('pcf_hip_knn_inverse_batched', [_I, _PP, _PP, _PP, _PP, _IP, _IP, _IP, _P, _Z, _P])
_gridsub_ws = _sig('pcf_hip_grid_subsample_workspace_bytes', [_I, _I], _Z)
_gridsub = _sig('pcf_hip_grid_subsample', [_P, _P, _P, _I, _I, _I, ctypes.c_float, _P, _P, _P, _P, _P, _Z, _P])
_vox_ws = _sig('pcf_hip_voxelize_workspace_bytes', [_I], _Z)
_vox = _sig('pcf_hip_voxelize', [_P, _I, ctypes.c_double, _I, ctypes.c_ulonglong, _I, _P, _P, _P, _Z, _P])
_vox_f64 = _sig('pcf_hip_voxelize_f64', [_P, _I, ctypes.c_double, _I, ctypes.c_ulonglong, _I, _P, _P, _P, _Z, _P])
_knn_grid = _sig('pcf_hip_knn_grid', [_P] * 4 + [_I] * 4 + [_P, _P, _Z, _P])
_knn_wave = _sig('pcf_hip_knn_wave', [_P] * 4 + [_I] * 3 + [_P] * 2)
_gemm_nt = _sig('pcf_hip_gemm_nt', [_P] * 4 + [_I] * 3 + [_P])


_log_enable = _sig('pcf_hip_launch_log_enable', [_I], None)
_log_read = _sig('pcf_hip_launch_log_read', [ctypes.c_char_p, _Z], _Z)
_set_engine = _sig('pcf_hip_set_aggregate_engine', [_I])
_get_engine = _sig('pcf_hip_get_aggregate_engine', [])

_set_chain_bwd_engine = _sig('pcf_hip_set_chain_backward_engine', [_I])

AGG_ENGINES = {'default': 0, 'lds': 1, 'tiled': 2, 'edge': 3}


def set_chain_backward_engine(lds_transposes: bool):
    """Last pass of the fused edge-graph backward: LDS transposes (True, default) or register layouts (False)."""
    if _set_chain_bwd_engine(1 if lds_transposes else 0) != 0:
        raise RuntimeError(_last_error().decode())


def library_path() -> str:
    return _LIB_PATH


def launch_log(enable: bool):
    """Start (clearing) or stop recording the names of the kernels the library launches (diagnostic / test hook)."""
    _log_enable(1 if enable else 0)


def read_launch_log():
    """-> list of kernel / launch-site names recorded since the last read."""
    n = _log_read(None, 0)
    buf = ctypes.create_string_buffer(n + 65536)
    _log_read(buf, len(buf))
    return [l for l in buf.value.decode().split('\n') if l]


def set_flin_finish(separate_launch: bool):
    """True (default): the statistics of the fused contraction kernels are combined by a launch of their own; False: by the
    last workgroup of the producing kernel."""
    _lib.pcf_hip_set_flin_finish(1 if separate_launch else 0)


def set_row_chain_finish(separate_launch: bool):
    """The same choice as set_flin_finish for the row chains (PCFLayer head / tail, pe_convs)."""
    _lib.pcf_hip_set_row_chain_finish(1 if separate_launch else 0)


def set_flin_split_k(mode: int):
    """-1: the split-K form of the point-level Linear+BN products where it pays (default); 0: never; 1: wherever it applies."""
    if _lib.pcf_hip_set_flin_split_k(int(mode)) != 0:
        raise RuntimeError(_last_error().decode())


def set_aggregate_engine(name: str):
    """'default' | 'lds' | 'tiled': the kernel family behind pcf_forward/backward and pconv_* for the shapes the
    matrix-core kernels cover (process-wide; the cross-checks of the test-suite toggle it).  'edge': additionally the
    thread-per-edge backward for unguided layers with C_mid 4 / 16 and <= 64 channels per edge (opt-in, unmeasured)."""
    if _set_engine(AGG_ENGINES[name]) != 0:
        raise RuntimeError(_last_error().decode())


def get_aggregate_engine() -> str:
    v = _get_engine()
    return next(k for k, e in AGG_ENGINES.items() if e == v)


def version() -> str:
    return _version().decode()


# ---- argument checks: same conditions and wording as CHECK_INPUT (pcf.h:14-24) -----------------
def _check_input(t, name, dtype=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor')
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA tensor')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f'{name} must be {dtype}, got {t.dtype}')


def _floats(**tensors):
    for name, t in tensors.items():
        _check_input(t, name, torch.float32)


def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


_raw_stream = torch._C._cuda_getCurrentRawStream
_exchange_device = torch.cuda._exchange_device
_maybe_exchange_device = torch.cuda._maybe_exchange_device


def _stream(dev):
    """torch's current HIP stream of `dev` as a raw handle (a training iteration asks ~400 times: the Stream object
    route costs 5 us per call, this one 0.3)."""
    return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())


class _guard:
    """`with torch.cuda.device(dev)` without its Python-side index parsing (2 us per use, ~450 uses per iteration)."""
    __slots__ = ('idx', 'prev')

    def __init__(self, dev):
        self.idx = -1 if dev.index is None else dev.index

    def __enter__(self):
        self.prev = _exchange_device(self.idx)

    def __exit__(self, *exc):
        _maybe_exchange_device(self.prev)
        return False


_timeline = None   # when a list: (entry point name, start event, end event) per bracketed C-ABI call
_timeline_only = None


def record_kernel_times(enable: bool, only=None):
    """Bracket C-ABI calls with HIP events on the stream they are enqueued on (torch's current stream);
    returns the list being filled, or None when disabled.  ``only`` restricts the bracketing to a set of
    entry-point names: an event pair costs the host several microseconds, so bench.py brackets just the
    dominant kernel inside the timed region and everything in a separate, untimed pass."""
    global _timeline, _timeline_only
    _timeline = [] if enable else None
    _timeline_only = set(only) if (enable and only) else None
    return _timeline


_TRACE_CALLS = os.environ.get('PCF_TRACE_CALLS') == '1'      # debugging aid: name every entry point on stderr before it runs


def _call(fn, *args):
    if _TRACE_CALLS:
        import sys
        print('pcf_cuda call:', fn.__name__, [a for a in args if isinstance(a, int) and abs(a) < (1 << 40)][:14], file=sys.stderr, flush=True)
    if _timeline is None or (_timeline_only is not None and fn.__name__ not in _timeline_only):
        rc = fn(*args)
    else:
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        rc = fn(*args)
        t1.record()
        _timeline.append((fn.__name__, t0, t1))
    if rc != 0:
        raise RuntimeError(f'pcf_cuda: {_last_error().decode()} (code {rc})')


def _dims(input, neighbor_inds, weights):
    if input.dim() != 3 or neighbor_inds.dim() != 3 or weights.dim() != 4:
        raise RuntimeError('pcf_cuda: expected input [B,N,C], neighbor_inds [B,Nout,K], weights [B,Nout,K,C_mid]')
    B, N, Ci = input.shape
    Bn, Nout, K = neighbor_inds.shape
    if Bn != B or tuple(weights.shape[:3]) != (B, Nout, K):
        raise RuntimeError('pcf_cuda: batch / point / neighbour dimensions of the arguments disagree')
    return B, N, Nout, K, Ci, weights.shape[3]


def _same_device(*ts):
    dev = ts[0].device
    for t in ts:
        if t is not None and t.device != dev:
            raise RuntimeError('pcf_cuda: all tensors must live on the same device')
    return dev


# ---- pcf_forward / pcf_backward  (pcf_cuda.cpp:10-11) -------------------------------------------
def pcf_forward(input, neighbor_inds, guidance, weights):
    _floats(input=input, guidance=guidance, weights=weights)
    _check_input(neighbor_inds, 'neighbor_inds', torch.int64)
    B, N, Nout, K, Ci, Cm = _dims(input, neighbor_inds, weights)
    if guidance.dim() != 4 or tuple(guidance.shape[:3]) != (B, Nout, K):
        raise RuntimeError('pcf_cuda: guidance must be [B,Nout,K,num_heads]')
    H = guidance.shape[3]
    dev = _same_device(input, neighbor_inds, guidance, weights)
    out = torch.empty(B, Nout, Ci * Cm, dtype=torch.float32, device=dev)
    with _guard(dev):
        _call(_pcf_fwd, _ptr(input), _ptr(neighbor_inds), _ptr(guidance), _ptr(weights), _ptr(out),
              B, N, Nout, K, Ci, Cm, H, _stream(dev))
    return out


def pcf_backward(grad_output, input, neighbor_inds, guidance, weights):
    # the reference forgets CHECK_INPUT(input) here (src/pcf.cu:33-36); checking it is harmless
    _floats(grad_output=grad_output, input=input, guidance=guidance, weights=weights)
    _check_input(neighbor_inds, 'neighbor_inds', torch.int64)
    B, N, Nout, K, Ci, Cm = _dims(input, neighbor_inds, weights)
    H = guidance.shape[3]
    if grad_output.numel() != B * Nout * Ci * Cm:
        raise RuntimeError('pcf_cuda: grad_output must be [B,Nout,C_in*C_mid]')
    dev = _same_device(grad_output, input, neighbor_inds, guidance, weights)
    grad_input = torch.empty_like(input)
    grad_guidance = torch.empty_like(guidance)
    grad_weights = torch.empty_like(weights)
    with _guard(dev):
        _call(_pcf_bwd, _ptr(grad_output), _ptr(input), _ptr(neighbor_inds), _ptr(guidance), _ptr(weights),
              _ptr(grad_input), _ptr(grad_guidance), _ptr(grad_weights), B, N, Nout, K, Ci, Cm, H, _stream(dev))
    return [grad_input, grad_guidance, grad_weights]


def pcf_backward_csr(grad_output, input, inverse_neighbor, inverse_neighbor_k, inverse_neighbor_idx, neighbor_inds,
                     guidance, weights):
    """pcf_backward with a deterministic, atomic-free grad_input: gather-reduce over the inverse CSR
    that compute_knn_inverse built for `neighbor_inds` (not in the reference's module)."""
    _floats(grad_output=grad_output, input=input, guidance=guidance, weights=weights)
    _check_input(neighbor_inds, 'neighbor_inds', torch.int64)
    _check_input(inverse_neighbor, 'inverse_neighbor', torch.int32)
    _check_input(inverse_neighbor_k, 'inverse_neighbor_k', torch.uint8)
    _check_input(inverse_neighbor_idx, 'inverse_neighbor_idx', torch.int32)
    B, N, Nout, K, Ci, Cm = _dims(input, neighbor_inds, weights)
    H = guidance.shape[3]
    if grad_output.numel() != B * Nout * Ci * Cm:
        raise RuntimeError('pcf_cuda: grad_output must be [B,Nout,C_in*C_mid]')
    dev = _same_device(grad_output, input, neighbor_inds, guidance, weights, inverse_neighbor)
    grad_input = torch.empty_like(input)
    grad_guidance = torch.empty_like(guidance)
    grad_weights = torch.empty_like(weights)
    nbytes = _pcf_bwd_csr_ws(B, Nout, K, Ci)
    ws = _workspace(nbytes, dev)
    with _guard(dev):
        _call(_pcf_bwd_csr, _ptr(grad_output), _ptr(input), _ptr(inverse_neighbor), _ptr(inverse_neighbor_k),
              _ptr(inverse_neighbor_idx), _ptr(neighbor_inds), _ptr(guidance), _ptr(weights), _ptr(grad_input),
              _ptr(grad_guidance), _ptr(grad_weights), ws.data_ptr(), nbytes, B, N, Nout, K, Ci, Cm, H,
              inverse_neighbor.shape[1], inverse_neighbor_idx.shape[1], _stream(dev))
    return [grad_input, grad_guidance, grad_weights]


# ---- pconv_forward / pconv_backward  (pcf_cuda.cpp:12,14) -------------------------------------
def _add_dims(additional_features, B, Nout, K):
    if additional_features.dim() != 4 or tuple(additional_features.shape[:3]) != (B, Nout, K):
        raise RuntimeError('pcf_cuda: additional_features must be [B,Nout,K,C_add] (C_add may be 0)')
    return additional_features.shape[3]


def pconv_forward(input, neighbor_inds, weights, additional_features):
    _floats(input=input, weights=weights, additional_features=additional_features)
    _check_input(neighbor_inds, 'neighbor_inds', torch.int64)
    B, N, Nout, K, Ci, Cm = _dims(input, neighbor_inds, weights)
    Ca = _add_dims(additional_features, B, Nout, K)
    dev = _same_device(input, neighbor_inds, weights, additional_features)
    out = torch.empty(B, Nout, (Ci + Ca) * Cm, dtype=torch.float32, device=dev)
    with _guard(dev):
        _call(_pconv_fwd, _ptr(input), _ptr(neighbor_inds), _ptr(weights), _ptr(additional_features), _ptr(out),
              B, N, Nout, K, Ci, Ca, Cm, _stream(dev))
    return out


def pconv_backward(grad_output, input, neighbor_inds, weights, additional_features):
    _floats(grad_output=grad_output, input=input, weights=weights, additional_features=additional_features)
    _check_input(neighbor_inds, 'neighbor_inds', torch.int64)
    B, N, Nout, K, Ci, Cm = _dims(input, neighbor_inds, weights)
    Ca = _add_dims(additional_features, B, Nout, K)
    if grad_output.numel() != B * Nout * (Ci + Ca) * Cm:
        raise RuntimeError('pcf_cuda: grad_output must be [B,Nout,(C_in+C_add)*C_mid]')
    dev = _same_device(grad_output, input, neighbor_inds, weights, additional_features)
    grad_input = torch.empty_like(input)
    grad_weights = torch.empty_like(weights)
    grad_additional = torch.empty_like(additional_features)
    with _guard(dev):
        _call(_pconv_bwd, _ptr(grad_output), _ptr(input), _ptr(neighbor_inds), _ptr(weights),
              _ptr(additional_features), _ptr(grad_input), _ptr(grad_weights), _ptr(grad_additional),
              B, N, Nout, K, Ci, Ca, Cm, _stream(dev))
    return [grad_input, grad_weights, grad_additional]


# ---- fused aggregate + linear  (pcf_cuda.cpp:13,15,16,18) --------------------------------------
def _linear_dims(linear_weights, Ci, Ca, Cm):
    if linear_weights.dim() != 2 or linear_weights.shape[1] != (Ci + Ca) * Cm:
        raise RuntimeError('pcf_cuda: linear_weights must be [C_out, (C_in+C_add)*C_mid]')
    return linear_weights.shape[0]


def pconv_linear_forward(input, neighbor_inds, weights, additional_features, linear_weights, linear_bias):
    _floats(input=input, weights=weights, additional_features=additional_features,
            linear_weights=linear_weights, linear_bias=linear_bias)
    _check_input(neighbor_inds, 'neighbor_inds', torch.int64)
    B, N, Nout, K, Ci, Cm = _dims(input, neighbor_inds, weights)
    Ca = _add_dims(additional_features, B, Nout, K)
    Co = _linear_dims(linear_weights, Ci, Ca, Cm)
    if linear_bias.numel() != Co:
        raise RuntimeError('pcf_cuda: linear_bias must be [C_out]')
    dev = _same_device(input, neighbor_inds, weights, additional_features, linear_weights, linear_bias)
    out = torch.empty(B, Nout, Co, dtype=torch.float32, device=dev)
    pconv_out = torch.empty(B, Nout, (Ci + Ca) * Cm, dtype=torch.float32, device=dev)
    with _guard(dev):
        _call(_pl_fwd, _ptr(input), _ptr(neighbor_inds), _ptr(weights), _ptr(additional_features),
              _ptr(linear_weights), _ptr(linear_bias), _ptr(out), _ptr(pconv_out),
              B, N, Nout, K, Ci, Ca, Cm, Co, _stream(dev))
    return [out, pconv_out]


# The reference exports a second forward built on CUTLASS GEMMs; on MI355X both names are the
# same MFMA path.
pconv_linear_cutlass_forward = pconv_linear_forward


def _workspace(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


def pconv_linear_backward(grad_output, input, neighbor_inds, weights, additional_features, linear_weights,
                          pconv_output):
    _floats(grad_output=grad_output, input=input, weights=weights, additional_features=additional_features,
            linear_weights=linear_weights, pconv_output=pconv_output)
    _check_input(neighbor_inds, 'neighbor_inds', torch.int64)
    B, N, Nout, K, Ci, Cm = _dims(input, neighbor_inds, weights)
    Ca = _add_dims(additional_features, B, Nout, K)
    Co = _linear_dims(linear_weights, Ci, Ca, Cm)
    if grad_output.numel() != B * Nout * Co or pconv_output.numel() != B * Nout * (Ci + Ca) * Cm:
        raise RuntimeError('pcf_cuda: grad_output must be [B,Nout,C_out] and pconv_output [B,Nout,(C_in+C_add)*C_mid]')
    dev = _same_device(grad_output, input, neighbor_inds, weights, additional_features, linear_weights, pconv_output)
    grad_input = torch.empty_like(input)
    grad_weights = torch.empty_like(weights)
    grad_additional = torch.empty_like(additional_features)
    grad_linear_weights = torch.empty_like(linear_weights)
    grad_linear_bias = torch.empty(Co, dtype=torch.float32, device=dev)
    nbytes = _pl_bwd_ws(B, N, Nout, K, Ci, Ca, Cm, Co)
    ws = _workspace(nbytes, dev)
    with _guard(dev):
        _call(_pl_bwd, _ptr(grad_output), _ptr(input), _ptr(neighbor_inds), _ptr(weights), _ptr(additional_features),
              _ptr(linear_weights), _ptr(pconv_output), _ptr(grad_input), _ptr(grad_weights), _ptr(grad_additional),
              _ptr(grad_linear_weights), _ptr(grad_linear_bias), ws.data_ptr(), nbytes,
              B, N, Nout, K, Ci, Ca, Cm, Co, _stream(dev))
    return [grad_input, grad_weights, grad_additional, grad_linear_weights, grad_linear_bias]


def pconv_linear_opt_backward(grad_output, input, inverse_neighbor, inverse_neighbor_k, inverse_neighbor_idx,
                              neighbor_inds, weights, additional_features, linear_weights, pconv_output):
    _floats(grad_output=grad_output, input=input, weights=weights, additional_features=additional_features,
            linear_weights=linear_weights, pconv_output=pconv_output)
    _check_input(neighbor_inds, 'neighbor_inds', torch.int64)
    _check_input(inverse_neighbor, 'inverse_neighbor', torch.int32)
    _check_input(inverse_neighbor_k, 'inverse_neighbor_k', torch.uint8)
    _check_input(inverse_neighbor_idx, 'inverse_neighbor_idx', torch.int32)
    B, N, Nout, K, Ci, Cm = _dims(input, neighbor_inds, weights)
    Ca = _add_dims(additional_features, B, Nout, K)
    Co = _linear_dims(linear_weights, Ci, Ca, Cm)
    if grad_output.numel() != B * Nout * Co or pconv_output.numel() != B * Nout * (Ci + Ca) * Cm:
        raise RuntimeError('pcf_cuda: grad_output must be [B,Nout,C_out] and pconv_output [B,Nout,(C_in+C_add)*C_mid]')
    if inverse_neighbor.dim() != 2 or inverse_neighbor_k.shape != inverse_neighbor.shape \
            or inverse_neighbor_idx.dim() != 2 or inverse_neighbor.shape[0] != B or inverse_neighbor_idx.shape[0] != B:
        raise RuntimeError('pcf_cuda: inverse index tensors must be [B,L], [B,L] and [B,>=N+1]')
    inv_len, inv_idx_len = inverse_neighbor.shape[1], inverse_neighbor_idx.shape[1]
    dev = _same_device(grad_output, input, inverse_neighbor, inverse_neighbor_k, inverse_neighbor_idx, neighbor_inds,
                       weights, additional_features, linear_weights, pconv_output)
    grad_input = torch.empty_like(input)
    grad_weights = torch.empty_like(weights)
    grad_additional = torch.empty_like(additional_features)
    grad_linear_weights = torch.empty_like(linear_weights)
    grad_linear_bias = torch.empty(Co, dtype=torch.float32, device=dev)
    nbytes = _plo_bwd_ws(B, N, Nout, K, Ci, Ca, Cm, Co)
    ws = _workspace(nbytes, dev)
    with _guard(dev):
        _call(_plo_bwd, _ptr(grad_output), _ptr(input), _ptr(inverse_neighbor), _ptr(inverse_neighbor_k),
              _ptr(inverse_neighbor_idx), _ptr(neighbor_inds), _ptr(weights), _ptr(additional_features),
              _ptr(linear_weights), _ptr(pconv_output), _ptr(grad_input), _ptr(grad_weights), _ptr(grad_additional),
              _ptr(grad_linear_weights), _ptr(grad_linear_bias), ws.data_ptr(), nbytes,
              B, N, Nout, K, Ci, Ca, Cm, Co, inv_len, inv_idx_len, _stream(dev))
    return [grad_input, grad_weights, grad_additional, grad_linear_weights, grad_linear_bias]


# ---- compute_knn_inverse  (pcf_cuda.cpp:17; src/pcf.cu:124-131 checks device + contiguity only) --
def compute_knn_inverse(neighbor_inds, total_points):
    _check_input(neighbor_inds, 'neighbor_inds', torch.int64)
    if neighbor_inds.dim() != 3:
        raise RuntimeError('pcf_cuda: neighbor_inds must be [B,N,K]')
    B, Nq, K = neighbor_inds.shape
    total_points = int(total_points)
    dev = neighbor_inds.device
    inv_neighbors = torch.empty(B, Nq * K, dtype=torch.int32, device=dev)
    inv_k = torch.empty(B, Nq * K, dtype=torch.uint8, device=dev)
    inv_idx = torch.empty(B, total_points + 1, dtype=torch.int32, device=dev)
    nbytes = _inv_ws(B, Nq, K, total_points)
    ws = _workspace(nbytes, dev)
    with _guard(dev):
        _call(_inv, _ptr(neighbor_inds), _ptr(inv_neighbors), _ptr(inv_k), inv_idx.data_ptr(), ws.data_ptr(), nbytes,
              B, Nq, K, total_points, _stream(dev))
    return [inv_neighbors, inv_k, inv_idx]


def compute_knn_inverse_batched(neighbor_inds_list, total_points_list):
    """compute_knn_inverse for several neighbour tables at once: every phase of the transpose runs ONE launch over all
    tables (the training loop transposes 3 x levels tables per iteration, util/common_util.py:281-309).  Tables are
    [1, Nq, K] int64 on one device; returns a list of [inv_neighbors, inv_k, inv_idx] with the contents and shapes
    compute_knn_inverse gives for each.  Empty tables take the single-table path."""
    tables = list(neighbor_inds_list)
    totals = [int(t) for t in total_points_list]
    if len(tables) != len(totals):
        raise RuntimeError('pcf_cuda: compute_knn_inverse_batched needs one total_points per table')
    out = [None] * len(tables)
    live = []
    for i, (t, tp) in enumerate(zip(tables, totals)):
        _check_input(t, 'neighbor_inds', torch.int64)
        if t.dim() != 3:
            raise RuntimeError('pcf_cuda: neighbor_inds must be [B,N,K]')
        if t.shape[0] != 1 or t.shape[1] == 0 or tp == 0:
            out[i] = compute_knn_inverse(t, tp)
        else:
            live.append(i)
    if not live:
        return out
    dev = _same_device(*[tables[i] for i in live])
    n = len(live)
    Nq = (ctypes.c_int * n)(*[tables[i].shape[1] for i in live])
    K = (ctypes.c_int * n)(*[tables[i].shape[2] for i in live])
    TP = (ctypes.c_int * n)(*[totals[i] for i in live])
    idx_p, n_p, k_p, x_p = ((ctypes.c_void_p * n)() for _ in range(4))
    for j, i in enumerate(live):
        t = tables[i]
        e = t.shape[1] * t.shape[2]
        trio = [torch.empty(1, e, dtype=torch.int32, device=dev), torch.empty(1, e, dtype=torch.uint8, device=dev),
                torch.empty(1, totals[i] + 1, dtype=torch.int32, device=dev)]
        out[i] = trio
        idx_p[j], n_p[j], k_p[j], x_p[j] = t.data_ptr(), trio[0].data_ptr(), trio[1].data_ptr(), trio[2].data_ptr()
    nbytes = _inv_batch_ws(n, Nq, K, TP)
    ws = _workspace(nbytes, dev)
    with _guard(dev):
        _call(_inv_batch, n, idx_p, n_p, k_p, x_p, Nq, K, TP, ws.data_ptr(), nbytes, _stream(dev))
    return out


# ---- extras beyond the reference's nine (used by knn_post_dataloader_utils and the tests) --------
KNN_GRID_MIN_REFS = 2048     # below this the brute-force kernel is as fast and needs no index
KNN_WAVE_MAX_WORK = 20_000   # queries x ceil(refs per sample / 1024): up to here a wave per query beats both engines
                             # (3.9k queries x 1k refs: 42 us vs 110 grid; 1.3k x 320: 15 vs 223 brute; 14.5k x 3.6k: 160 vs 119 grid)


def knn_packed(ref, query, ref_offsets, query_offsets, K, method='auto'):
    """K nearest refs (own sample only) for every query of a packed batch.

    ref [Nr,3] f32, query [Nq,3] f32 device tensors; ref_offsets / query_offsets int32 [S+1]
    device tensors of per-sample prefix offsets.  Returns int64 [Nq,K] of packed ref indices,
    (distance, index) ascending; -1 where a sample has fewer than K refs.  ``method``: 'brute'
    (tiled brute force, a lane per query), 'grid' (uniform-grid index), 'wave' (brute force, a wave per query: the
    coarse levels) or 'auto' (wave for small problems, else grid from 2048 reference points up); all engines
    return bit-identical results."""
    _floats(ref=ref, query=query)
    _check_input(ref_offsets, 'ref_offsets', torch.int32)
    _check_input(query_offsets, 'query_offsets', torch.int32)
    if ref.dim() != 2 or ref.shape[1] != 3 or query.dim() != 2 or query.shape[1] != 3:
        raise RuntimeError('pcf_cuda: ref and query must be [n,3]')
    S = ref_offsets.numel() - 1
    if S < 0 or query_offsets.numel() != S + 1:
        raise RuntimeError('pcf_cuda: offset tensors must both be [num_samples+1]')
    dev = _same_device(ref, query, ref_offsets, query_offsets)
    out = torch.empty(query.shape[0], K, dtype=torch.int64, device=dev)
    if method not in ('auto', 'brute', 'grid', 'wave'):
        raise ValueError(f'knn_packed: unknown method {method!r}')
    n_ref, n_query = ref.shape[0], query.shape[0]
    chunks = -(-n_ref // (max(S, 1) * 1024))             # per-sample sizes are device data: use the batch average
    use_wave = method == 'wave' or (method == 'auto' and n_query * max(chunks, 1) <= KNN_WAVE_MAX_WORK)
    use_grid = method == 'grid' or (method == 'auto' and not use_wave and n_ref >= KNN_GRID_MIN_REFS)
    with _guard(dev):
        if use_wave:
            _call(_knn_wave, _ptr(ref), _ptr(query), ref_offsets.data_ptr(), query_offsets.data_ptr(), S, n_query, int(K),
                  _ptr(out), _stream(dev))
        elif use_grid:
            nbytes = _knn_grid_ws(n_ref, S)
            ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
            _call(_knn_grid, _ptr(ref), _ptr(query), ref_offsets.data_ptr(), query_offsets.data_ptr(), S, n_ref, n_query,
                  int(K), _ptr(out), ws.data_ptr(), nbytes, _stream(dev))
        else:
            # the widest sample bounds the launch grid; a host-side upper bound avoids a device->host sync
            _call(_knn, _ptr(ref), _ptr(query), ref_offsets.data_ptr(), query_offsets.data_ptr(), S, n_query, int(K),
                  _ptr(out), _stream(dev))
    return out


_VOX_MODES = {'deterministic': 0, 'random': 1, 'rank': 2}


def voxelize(points, voxel_size, mode='deterministic', seed=0, rank=0):
    """At most one point per occupied voxel (util/voxelize.py:44-82 on the GPU, FNV hash).  points [N,3] f32 device
    tensor of ONE cloud -> (idx int64 [V] in ascending key order, fullest-voxel count).  mode 'deterministic': the voxel's
    lowest point index; 'random': a pseudo-random point of the voxel from `seed`; 'rank': point `rank` mod count.  Reading
    V back is the one device->host sync (data-dependent output size).  float64 points are hashed from their double values
    (pcf_hip_voxelize_f64), as numpy does for a float64 coordinate array."""
    f64 = isinstance(points, torch.Tensor) and points.dtype == torch.float64
    _check_input(points, 'points', torch.float64 if f64 else torch.float32)
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError('pcf_cuda: points must be [n,3]')
    n = points.shape[0]
    dev = points.device
    out = torch.empty(n, dtype=torch.int64, device=dev)
    meta = torch.empty(2, dtype=torch.int32, device=dev)
    with _guard(dev):
        nbytes = _vox_ws(n)
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
        _call(_vox_f64 if f64 else _vox, _ptr(points), n, float(voxel_size), _VOX_MODES[mode], int(seed) & (2 ** 64 - 1), int(rank), out.data_ptr(),
              meta.data_ptr(), ws.data_ptr(), nbytes, _stream(dev))
    total, longest = meta.cpu().tolist()
    return out[:total], longest


def grid_subsample(points, features=None, offsets=None, sampleDl=0.1):
    """Barycentre grid subsampling of a packed batch (GPU form of ``cpp_subsampling.compute(points, features=...,
    sampleDl=..., method="barycenters")``, datasetCommon.py:17-67, applied to every sample of the batch in one call).

    points [N,3] f32 and optional features [N,F] f32 device tensors packed over S samples; offsets int32 [S+1]
    device tensor of per-sample prefix offsets (None: one sample).  Returns (sub_points [M,3], sub_features [M,F] or
    None, counts) with ``counts`` a host list of the voxels per sample: voxels are ordered by sample, then by the
    reference's linear voxel index; barycentres and feature means are bit-identical to the reference's.  Reading
    ``counts`` is the one device->host sync (the output size is data dependent, as with torch.unique)."""
    _floats(points=points)
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError('pcf_cuda: points must be [n,3]')
    n = points.shape[0]
    F = 0
    if features is not None:
        _floats(features=features)
        if features.dim() != 2 or features.shape[0] != n:
            raise RuntimeError('pcf_cuda: features must be [n,F] with one row per point')
        F = features.shape[1]
    dev = points.device
    if offsets is None:
        offsets = torch.tensor([0, n], dtype=torch.int32, device=dev)
    _check_input(offsets, 'offsets', torch.int32)
    S = offsets.numel() - 1
    if S < 0:
        raise RuntimeError('pcf_cuda: offsets must be [num_samples+1]')
    if not sampleDl > 0:
        raise ValueError('grid_subsample: sampleDl must be positive')
    out_p = torch.empty(n, 3, dtype=torch.float32, device=dev)
    out_f = torch.empty(n, F, dtype=torch.float32, device=dev) if F else None
    meta = torch.empty(S + 2, dtype=torch.int32, device=dev)          # [total, status, counts...]
    with _guard(dev):
        nbytes = _gridsub_ws(n, S)
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
        _call(_gridsub, _ptr(points), _ptr(features) if F else None, offsets.data_ptr(), S, n, F, float(sampleDl),
              _ptr(out_p), _ptr(out_f) if F else None, meta.data_ptr() + 8, meta.data_ptr(), ws.data_ptr(), nbytes,
              _stream(dev))
    host = meta.cpu().tolist()
    total, status, counts = host[0], host[1], host[2:]
    if status != 0:
        raise RuntimeError('pcf_cuda: grid_subsample: a sample spans 2^18 or more voxels along an axis (or holds '
                           'non-finite coordinates) at sampleDl=%g' % sampleDl)
    return out_p[:total], (out_f[:total] if F else None), counts


def gemm_nt(a, b, bias=None):
    """a [M,Kd] . b[N,Kd]^T (+ bias[N]) on the fp32 MFMA path (exposed for tests / roofline)."""
    _floats(a=a, b=b)
    if bias is not None:
        _floats(bias=bias)
    M, Kd = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    with _guard(a.device):
        _call(_gemm_nt, _ptr(a), _ptr(b), _ptr(bias), _ptr(out), M, N, Kd, _stream(a.device))
    return out
