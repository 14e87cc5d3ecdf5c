"""CPU: the C-ABI library loads and exports every symbol include/pcf_hip.h declares; the Python
`pcf_cuda` module exposes the reference's nine functions and rejects host tensors the way the
reference's CHECK_INPUT does.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'pcf_hip.h')).read()
    return sorted(set(re.findall(r'\b(pcf_hip_\w+)\s*\(', text)))


def test_header_symbols_exported():
    import pcf_cuda
    lib = ctypes.CDLL(pcf_cuda.library_path())
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/pcf_hip.h but not exported'


def test_every_export_is_declared():
    """The reverse direction: no `pcf_hip_*` entry point ships without a declaration (and a reference citation)
    in the header."""
    import shutil
    import subprocess
    import pcf_cuda
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    if not os.path.exists(nm):
        pytest.skip('no nm in this image')
    out = subprocess.run([nm, '-D', '--defined-only', pcf_cuda.library_path()], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-1].startswith('pcf_hip_')})
    assert len(exported) >= 40
    missing = [n for n in exported if n not in _declared()]
    assert not missing, f'exported but not declared in include/pcf_hip.h: {missing}'


def test_module_surface_matches_reference():
    import pcf_cuda
    # pcf_cuda.cpp:10-18
    for n in ['pcf_forward', 'pcf_backward', 'pconv_forward', 'pconv_linear_forward', 'pconv_backward',
              'pconv_linear_backward', 'pconv_linear_opt_backward', 'compute_knn_inverse',
              'pconv_linear_cutlass_forward']:
        assert callable(getattr(pcf_cuda, n)), n
    assert 'gfx950' in pcf_cuda.version()


def test_host_tensors_rejected_like_check_input():
    import pcf_cuda
    x = torch.zeros(1, 4, 4)
    idx = torch.zeros(1, 4, 2, dtype=torch.long)
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        pcf_cuda.pcf_forward(x, idx, torch.zeros(1, 4, 2, 2), torch.zeros(1, 4, 2, 4))
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        pcf_cuda.compute_knn_inverse(idx, 4)


def test_c_abi_argument_errors_without_gpu():
    """Argument validation happens before any HIP call, so it can be exercised on CPU."""
    import pcf_cuda
    lib = ctypes.CDLL(pcf_cuda.library_path())
    lib.pcf_hip_last_error.restype = ctypes.c_char_p
    rc = lib.pcf_hip_knn(None, None, None, None, 1, 10, 65, None, None)
    assert rc == -1 and b'K must be in [1,64]' in lib.pcf_hip_last_error()
    lib.pcf_hip_knn_inverse_workspace_bytes.restype = ctypes.c_size_t
    assert lib.pcf_hip_knn_inverse_workspace_bytes(1, 100, 16, 100) > 100 * 16 * 4


def test_zero_kernel_indexing_on_host():
    """csrc/pcf_common.h:zero_item -- the per-thread share of the library's zero-fill kernel (which replaces every
    hipMemsetAsync, so that a captured iteration holds no memset nodes) -- run on host memory: every byte of [p, p+n) is
    cleared and not one byte outside, for all alignments of p, sizes around the 16-byte body granularity, one and many
    emulated workgroups."""
    import numpy as np
    import pcf_cuda
    lib = ctypes.CDLL(pcf_cuda.library_path())
    lib.pcf_hip_zero_host.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    lib.pcf_hip_zero_host.restype = None
    sizes = list(range(0, 70)) + [255, 256, 257, 4095, 4096, 4097, 86236, 156928, 1 << 20]
    for blocks in (1, 3, 64):
        for n in sizes:
            for shift in (0, 1, 3, 4, 8, 15):
                buf = np.full(n + 64, 0xAB, np.uint8)
                base = buf.ctypes.data
                start = (-base) % 16 + 16 + shift                # buf[start] sits `shift` bytes past a 16-byte boundary
                lib.pcf_hip_zero_host(ctypes.c_void_p(base + start), n, blocks)
                assert not buf[start:start + n].any(), (blocks, n, shift)
                assert (buf[:start] == 0xAB).all() and (buf[start + n:] == 0xAB).all(), (blocks, n, shift)


def test_public_header_is_plain_c():
    """include/pcf_hip.h is the drop-in boundary: it must compile on its own as C99 and as C++ (no torch, no HIP types)."""
    import shutil
    import subprocess
    header = os.path.join(ROOT, 'include', 'pcf_hip.h')
    for cc, lang in (('gcc', ['-x', 'c', '-std=c99']), ('g++', ['-x', 'c++'])):
        if shutil.which(cc) is None:
            pytest.skip(f'no {cc} in this image')
        res = subprocess.run([cc, '-fsyntax-only', '-Wall'] + lang + [header], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr


@pytest.mark.parametrize('B,N,Nout,K,Ci,Ca,Cm', [(1, 300, 300, 16, 6, 12, 16), (2, 90, 41, 8, 3, 0, 16), (1, 64, 64, 16, 16, 16, 4),
                                               (1, 50, 70, 5, 7, 5, 16), (1, 33, 33, 16, 1, 12, 4)])
def test_thread_per_edge_backward_body_against_oracle(B, N, Nout, K, Ci, Ca, Cm):
    """csrc/aggregate.hip:agg_bwd_edge_item -- the per-edge body of the opt-in thread-per-edge backward (engine 3) -- is
    __host__ __device__: pcf_hip_pconv_backward_edge_host runs it over every edge on host memory.  Against
    oracle.pcf_oracle.pconv_backward (autograd adjoint of layers.py:890-897): grad_w, grad_add, and grad_x both as an
    accumulation and as per-edge contribution rows summed over the edges; aligned and misaligned buffers (both load paths)."""
    import numpy as np
    import pcf_cuda
    from oracle import pcf_oracle as O
    lib = ctypes.CDLL(pcf_cuda.library_path())
    fn = lib.pcf_hip_pconv_backward_edge_host
    fn.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_int] * 8
    fn.restype = ctypes.c_int
    g = torch.Generator().manual_seed(B * 1000 + N + Ci)
    x = torch.randn(B, N, Ci, generator=g)
    idx = torch.randint(0, N, (B, Nout, K), generator=g)
    w = torch.randn(B, Nout, K, Cm, generator=g)
    add = torch.randn(B, Nout, K, Ca, generator=g)
    gout = torch.randn(B, Nout, (Ci + Ca) * Cm, generator=g)
    want_x, want_w, want_add = O.pconv_backward(gout, x, idx, w, add)

    def buf(t, shift):          # a copy of t at a 16-byte boundary (+ shift floats): aligned and misaligned runs
        store = np.zeros(t.numel() + 8, np.float32)
        off = (-store.ctypes.data // 4) % 4 + shift
        view = store[off:off + t.numel()]
        view[:] = t.reshape(-1).numpy()
        return store, view

    ptr = lambda v: ctypes.c_void_p(v.ctypes.data) if v.size else None
    for shift in (0, 1):
        keep = [buf(t, shift) for t in (gout, x, w, add)]
        (_, vg), (_, vx), (_, vw), (_, va) = keep
        idx_np = np.ascontiguousarray(idx.numpy())
        for atomic in (1, 0):
            gx = np.zeros(B * N * Ci, np.float32)
            contrib = np.zeros(B * Nout * K * Ci, np.float32)
            sgw, gw = buf(torch.zeros(B * Nout * K * Cm), shift)
            sga, ga = buf(torch.zeros(B * Nout * K * Ca), shift)
            rc = fn(ptr(vg), ptr(vx), ctypes.c_void_p(idx_np.ctypes.data), ptr(vw), ptr(va), ptr(gx), ptr(contrib), ptr(gw), ptr(ga),
                    B, N, Nout, K, Ci, Ca, Cm, atomic)
            assert rc == 0
            torch.testing.assert_close(torch.from_numpy(gw.copy()).reshape(want_w.shape), want_w, rtol=1e-5, atol=1e-5)
            if Ca:
                torch.testing.assert_close(torch.from_numpy(ga.copy()).reshape(want_add.shape), want_add, rtol=1e-5, atol=1e-5)
            if Ci:
                if atomic:
                    got = torch.from_numpy(gx).reshape(B, N, Ci)
                else:
                    got = torch.zeros(B, N, Ci)
                    rows = torch.from_numpy(contrib).reshape(B, Nout * K, Ci)
                    for b in range(B):
                        got[b].index_add_(0, idx[b].reshape(-1), rows[b])
                torch.testing.assert_close(got, want_x, rtol=1e-4, atol=1e-4)
    # out-of-range neighbour indices contribute nothing and receive nothing (the library's rule for every aggregate)
    if Ci:
        bad = idx.clone()
        bad[:, ::3, 0] = -1
        bad[:, 1::3, 1] = N + 5
        valid = (bad >= 0) & (bad < N)
        safe = torch.where(valid, bad, torch.zeros_like(bad))
        xs = torch.cat([x, torch.zeros(B, 1, Ci)], 1)                       # row N = zeros for the invalid slots
        idx_o = torch.where(valid, safe, torch.full_like(bad, N))
        want_x2, want_w2, want_add2 = O.pconv_backward(gout, xs, idx_o, w, add)
        gx = np.zeros(B * N * Ci, np.float32)
        gw = np.zeros(B * Nout * K * Cm, np.float32)
        ga = np.zeros(max(B * Nout * K * Ca, 1), np.float32)
        bad_np = np.ascontiguousarray(bad.numpy())
        arr = lambda t: np.ascontiguousarray(t.reshape(-1).numpy())
        vg, vx, vw, va = arr(gout), arr(x), arr(w), arr(add)
        assert fn(ptr(vg), ptr(vx), ctypes.c_void_p(bad_np.ctypes.data), ptr(vw), ptr(va), ptr(gx), None, ptr(gw), ptr(ga) if Ca else None,
                  B, N, Nout, K, Ci, Ca, Cm, 1) == 0
        torch.testing.assert_close(torch.from_numpy(gx).reshape(B, N, Ci), want_x2[:, :N], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(torch.from_numpy(gw).reshape(want_w2.shape), want_w2, rtol=1e-5, atol=1e-5)
