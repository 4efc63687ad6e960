"""The C-ABI shared library loads and exports every symbol include/cyten_amd.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / 'include' / 'cyten_amd.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cyb_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_hot_path():
    syms = _declared_symbols()
    for need in ('cyb_gemm_plan_create', 'cyb_gemm_plan_run', 'cyb_gemm_grouped_f64', 'cyb_svd_batched_f64',
                 'cyb_qr_batched_f64', 'cyb_eigh_batched_f64', 'cyb_copy_strided_batched', 'cyb_ctx_create'):
        assert need in syms


def test_library_exports_every_declared_symbol():
    from cyten_amd import _lib, build
    path = build.build(verbose=False)
    lib = ctypes.CDLL(str(path))
    syms = _declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/cyten_amd.h but not exported'
    # the ctypes layer binds exactly the declared functions
    assert sorted(_lib.PROTOTYPES) == syms
    assert _lib.load().cyb_version() == 100


def test_struct_layouts_match_the_header():
    """ctypes mirrors of the descriptor structs have the C sizes (LP64)."""
    from cyten_amd import _lib
    assert ctypes.sizeof(_lib.GemmSeg) == 7 * 8
    assert ctypes.sizeof(_lib.GemmProb) == 8 * 4 + 4 * 2 + 8 * 2
    assert ctypes.sizeof(_lib.SvdDesc) == 9 * 8
    assert ctypes.sizeof(_lib.QrDesc) == 8 * 8 + 8
    assert ctypes.sizeof(_lib.EighDesc) == 6 * 8
    assert ctypes.sizeof(_lib.CopyDesc) == 8 * 2 + 4 * 2 + 8 * 8 * 3
    assert ctypes.sizeof(_lib.VecDesc) == 4 * 8
    assert ctypes.sizeof(_lib.MaskDesc) == 7 * 8


def test_missing_library_fails_loudly(tmp_path):
    from cyten_amd import _lib
    with pytest.raises(ImportError):
        _lib.load(tmp_path / 'libcyten_amd.so')


def test_backend_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a HIP device is present')
    from cyten_amd.block_backend import HipBlockBackend
    with pytest.raises(RuntimeError):
        HipBlockBackend('cuda:0')
