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
