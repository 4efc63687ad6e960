"""ctypes binding of the C-ABI in ``include/cyten_amd.h`` (libcyten_amd.so, HIP for gfx950).

The product path has NO CPU fallback: if the shared library is missing or fails to load, or if a
call returns a non-zero status, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / 'lib' / 'libcyten_amd.so'

CYB_MAX_NDIM = 8
CYB_SVD_SKIP_NULL_VECTORS = 1
CYB_SVD_EMBEDDED_COMPLEX = 2
CYB_EIGH_EMBEDDED_COMPLEX = 2

CYB_OK, CYB_ERR_INVALID, CYB_ERR_HIP, CYB_ERR_NOCONV, CYB_ERR_NOMEM, CYB_ERR_UNSUPPORTED = range(6)


class CybError(RuntimeError):
    """A C-ABI call failed. ``status`` holds the CYB_ERR_* code."""

    def __init__(self, status: int, msg: str):
        super().__init__(f'libcyten_amd error {status}: {msg}')
        self.status = status


class LinAlgError(CybError):
    """An iterative decomposition did not converge (numpy.linalg.LinAlgError analogue)."""


class GemmSeg(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('K', C.c_int64),
                ('a_rs', C.c_int64), ('a_cs', C.c_int64), ('b_rs', C.c_int64), ('b_cs', C.c_int64)]


class GemmProb(C.Structure):
    _fields_ = [('C', C.c_void_p), ('M', C.c_int64), ('N', C.c_int64), ('ldc', C.c_int64),
                ('seg_begin', C.c_int32), ('seg_end', C.c_int32), ('alpha', C.c_double), ('beta', C.c_double)]


class SvdDesc(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int64), ('m', C.c_int64), ('n', C.c_int64),
                ('U', C.c_void_p), ('ldu', C.c_int64), ('S', C.c_void_p),
                ('Vh', C.c_void_p), ('ldvh', C.c_int64)]


class QrDesc(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int64), ('m', C.c_int64), ('n', C.c_int64),
                ('Q', C.c_void_p), ('ldq', C.c_int64), ('R', C.c_void_p), ('ldr', C.c_int64),
                ('full', C.c_int32)]


class EighDesc(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int64), ('n', C.c_int64),
                ('W', C.c_void_p), ('V', C.c_void_p), ('ldv', C.c_int64)]


class CopyDesc(C.Structure):
    _fields_ = [('dst', C.c_void_p), ('src', C.c_void_p), ('ndim', C.c_int32), ('conj', C.c_int32),
                ('shape', C.c_int64 * CYB_MAX_NDIM), ('dst_strides', C.c_int64 * CYB_MAX_NDIM),
                ('src_strides', C.c_int64 * CYB_MAX_NDIM)]


class VecDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('out', C.c_void_p), ('n', C.c_int64)]


class ScaleAxisDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('f', C.c_void_p), ('out', C.c_void_p),
                ('outer', C.c_int64), ('axis', C.c_int64), ('inner', C.c_int64)]


class CExpandDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('rs', C.c_int64), ('cs', C.c_int64), ('K', C.c_int64), ('N', C.c_int64),
                ('dst', C.c_void_p)]


class TruncOpts(C.Structure):
    _fields_ = [('chi_max', C.c_int64), ('chi_min', C.c_int64), ('degeneracy_tol', C.c_double), ('trunc_cut', C.c_double),
                ('svd_min', C.c_double), ('has_svd_min', C.c_int32), ('minimize_error', C.c_int32)]


class LegDesc(C.Structure):
    _fields_ = [('n_sectors', C.c_int64), ('sectors', C.c_void_p), ('mults', C.c_void_p), ('sign', C.c_int32), ('pad', C.c_int32)]


class LincombDesc(C.Structure):
    _fields_ = [('dst', C.c_void_p), ('ndim', C.c_int32), ('accumulate', C.c_int32), ('term_begin', C.c_int32),
                ('term_end', C.c_int32), ('shape', C.c_int64 * CYB_MAX_NDIM), ('dst_strides', C.c_int64 * CYB_MAX_NDIM)]


class LincombTerm(C.Structure):
    _fields_ = [('src', C.c_void_p), ('coeff', C.c_double), ('src_strides', C.c_int64 * CYB_MAX_NDIM)]


class LincombTermC128(C.Structure):
    _fields_ = [('src', C.c_void_p), ('coeff_re', C.c_double), ('coeff_im', C.c_double), ('src_real', C.c_int32),
                ('reserved', C.c_int32), ('src_strides', C.c_int64 * CYB_MAX_NDIM)]


class MaskDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('out', C.c_void_p), ('idx', C.c_void_p),
                ('outer', C.c_int64), ('axis', C.c_int64), ('inner', C.c_int64), ('n_keep', C.c_int64)]


# numpy views of the descriptor structs (same layout: numpy derives the dtype from the ctypes Structure), for the
# vectorised marshalling of long block lists
import numpy as _np  # noqa: E402

GEMM_SEG_DTYPE = _np.dtype(GemmSeg)
GEMM_PROB_DTYPE = _np.dtype(GemmProb)
VEC_DTYPE = _np.dtype(VecDesc)
COPY_DTYPE = _np.dtype(CopyDesc)
SVD_DTYPE = _np.dtype(SvdDesc)
QR_DTYPE = _np.dtype(QrDesc)
EIGH_DTYPE = _np.dtype(EighDesc)
LINCOMB_DTYPE = _np.dtype(LincombDesc)
LINTERM_DTYPE = _np.dtype(LincombTerm)
LINTERM_C128_DTYPE = _np.dtype(LincombTermC128)
CEXPAND_DTYPE = _np.dtype(CExpandDesc)

_P = C.POINTER
_ctx = C.c_void_p
_vp = C.c_void_p

# name -> (argtypes); every function returns int status unless listed in _NON_STATUS.
PROTOTYPES = {
    'cyb_version': [],
    'cyb_last_error': [],
    'cyb_ctx_create': [_P(_ctx), C.c_int, _vp],
    'cyb_ctx_destroy': [_ctx],
    'cyb_ctx_set_stream': [_ctx, _vp],
    'cyb_ctx_sync': [_ctx],
    'cyb_device_info': [_ctx, _P(C.c_int), _P(C.c_int), _P(C.c_int64), C.c_char_p, C.c_int],
    'cyb_malloc': [_ctx, _P(_vp), C.c_size_t],
    'cyb_free': [_ctx, _vp],
    'cyb_memcpy_h2d': [_ctx, _vp, _vp, C.c_size_t],
    'cyb_memcpy_d2h': [_ctx, _vp, _vp, C.c_size_t],
    'cyb_memcpy_d2d': [_ctx, _vp, _vp, C.c_size_t],
    'cyb_memset': [_ctx, _vp, C.c_int, C.c_size_t],
    'cyb_event_create': [_P(_vp)],
    'cyb_ctx_time_next_gemm': [_ctx, _vp, _vp],
    'cyb_event_destroy': [_vp],
    'cyb_event_record': [_ctx, _vp],
    'cyb_event_elapsed_ms': [_vp, _vp, _P(C.c_float)],
    'cyb_gemm_plan_create': [_ctx, _P(_vp), _P(GemmProb), C.c_int64, _P(GemmSeg), C.c_int64],
    'cyb_gemm_plan_run': [_ctx, _vp],
    'cyb_gemm_plan_destroy': [_vp],
    'cyb_gemm_plan_info': [_vp, _P(C.c_double), _P(C.c_double), _P(C.c_int64), _P(C.c_int32)],
    'cyb_gemm_grouped_f64': [_ctx, _P(GemmProb), C.c_int64, _P(GemmSeg), C.c_int64],
    'cyb_gemm_grouped_enqueue_f64': [_ctx, _P(GemmProb), C.c_int64, _P(GemmSeg), C.c_int64],
    'cyb_mfma_f64_peak': [_ctx, C.c_int, C.c_int, _P(C.c_double), _P(C.c_double)],
    'cyb_svd_batched_f64': [_ctx, _P(SvdDesc), C.c_int64, _P(C.c_int32)],
    'cyb_svd_batched_ex_f64': [_ctx, _P(SvdDesc), C.c_int64, _P(C.c_int32), C.c_int32, _P(C.c_int32)],
    'cyb_compose_plan_create': [_vp, C.c_int32, _P(LegDesc), C.c_int32, _vp, C.c_int64, _P(LegDesc), C.c_int32, _vp, C.c_int64,
                                C.c_int32, _P(_vp)],
    'cyb_compose_plan_sizes': [_vp, _P(C.c_int64), _P(C.c_int64), _P(C.c_int64)],
    'cyb_compose_plan_get': [_vp, _vp, _vp, _vp, _vp, _vp, _P(C.c_double)],
    'cyb_compose_plan_enqueue_f64': [_ctx, _vp, _vp, _vp, _vp, C.c_int64, _vp, _P(C.c_double), _P(C.c_double)],
    'cyb_compose_plan_destroy': [_vp],
    'cyb_svd_batched_c128': [_ctx, _P(SvdDesc), C.c_int64, _P(C.c_int32)],
    'cyb_eigh_batched_c128': [_ctx, _P(EighDesc), C.c_int64, _P(C.c_int32)],
    'cyb_qr_batched_c128': [_ctx, _P(QrDesc), C.c_int64],
    'cyb_qr_batched_f64': [_ctx, _P(QrDesc), C.c_int64],
    'cyb_eigh_batched_f64': [_ctx, _P(EighDesc), C.c_int64, _P(C.c_int32)],
    'cyb_eigh_batched_ex_f64': [_ctx, _P(EighDesc), C.c_int64, _P(C.c_int32), C.c_int32],
    'cyb_copy_strided_batched': [_ctx, _P(CopyDesc), C.c_int64, C.c_int32],
    'cyb_dot_batched_f64': [_ctx, _P(VecDesc), C.c_int64, _vp],
    'cyb_dot_each_f64': [_ctx, _P(VecDesc), C.c_int64, _vp],
    'cyb_axpby_batched_f64': [_ctx, _P(VecDesc), C.c_int64, C.c_double, C.c_double],
    'cyb_maxabs_batched_f64': [_ctx, _P(VecDesc), C.c_int64, _vp],
    'cyb_binary_batched_f64': [_ctx, _P(VecDesc), C.c_int64, C.c_int32],
    'cyb_unary_batched_f64': [_ctx, _P(VecDesc), C.c_int64, C.c_int32],
    'cyb_scale_axis_batched_f64': [_ctx, _P(ScaleAxisDesc), C.c_int64],
    'cyb_mask_gather_batched_f64': [_ctx, _P(MaskDesc), C.c_int64],
    'cyb_mask_scatter_batched_f64': [_ctx, _P(MaskDesc), C.c_int64],
    'cyb_complex_expand_batched_f64': [_ctx, _P(CExpandDesc), C.c_int64],
    'cyb_elementwise_batched_c128': [_ctx, _P(VecDesc), C.c_int64, C.c_int32],
    'cyb_axpby_batched_c128': [_ctx, _P(VecDesc), C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double],
    'cyb_fill_f64': [_ctx, _vp, C.c_int64, C.c_double],
    'cyb_eye_f64': [_ctx, _vp, C.c_int64],
    'cyb_random_normal_f64': [_ctx, _vp, C.c_int64, C.c_uint64, C.c_double],
    'cyb_lincomb_strided_batched_f64': [_ctx, _P(LincombDesc), C.c_int64, _P(LincombTerm), C.c_int64],
    'cyb_lincomb_strided_batched_c128': [_ctx, _P(LincombDesc), C.c_int64, _P(LincombTermC128), C.c_int64],
    'cyb_truncate_select_f64': [_ctx, _P(VecDesc), C.c_int64, _P(TruncOpts), _vp, _vp, _vp],
    'cyb_truncate_select_weighted_f64': [_ctx, _P(VecDesc), C.c_int64, _vp, _P(TruncOpts), _vp, _vp, _vp],
    'cyb_random_uniform_f64': [_ctx, _vp, C.c_int64, C.c_uint64, C.c_double, C.c_double],
    'cyb_unary_param_batched_f64': [_ctx, _P(VecDesc), C.c_int64, C.c_int32, C.c_double],
    'cyb_compare_f64': [_ctx, _vp, _vp, C.c_double, _vp, C.c_int64, C.c_int32],
    'cyb_convert_u8_f64': [_ctx, _vp, _vp, C.c_int64],
    'cyb_count_nonzero_u8': [_ctx, _vp, C.c_int64, _vp],
    'cyb_extremum_f64': [_ctx, _vp, C.c_int64, C.c_int32, _vp],
}
_NON_STATUS = {'cyb_version': C.c_int, 'cyb_last_error': C.c_char_p}

_lib = None


def load(path: Path | None = None):
    """Load libcyten_amd.so and declare every prototype. Raises if the library is missing."""
    global _lib
    import os
    if _lib is not None and path is None:
        return _lib
    if path is None and os.environ.get('CYTEN_AMD_LIB'):  # A/B builds of the kernels (development only)
        path = os.environ['CYTEN_AMD_LIB']
    p = Path(path) if path is not None else LIB_PATH
    if not p.exists():
        raise ImportError(
            f'{p} not found: the HIP extension is not built. Run `python -m cyten_amd.build` '
            '(or __graft_entry__.build()). cyten_amd has no CPU fallback.')
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE
    # this library is loaded, otherwise libcyten_amd binds to the system copy under /opt/rocm and the
    # process ends up with two HIP runtimes (observed: "no ROCm-capable device is detected" from the
    # second one).  Importing torch first makes both use the same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(p))
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _NON_STATUS.get(name, C.c_int)
    if path is None or str(path) == os.environ.get('CYTEN_AMD_LIB'):
        _lib = lib
    return lib


def check(status: int):
    """Raise the exception matching a non-zero C-ABI status (error behaviour of the reference:
    invalid arguments -> ValueError (std::invalid_argument, numpy.cpp:1296), non-convergence ->
    LinAlgError, everything else RuntimeError)."""
    if status == CYB_OK:
        return
    msg = load().cyb_last_error().decode(errors='replace')
    if status == CYB_ERR_INVALID:
        raise ValueError(f'libcyten_amd: {msg}')
    if status == CYB_ERR_NOCONV:
        raise LinAlgError(status, msg)
    if status == CYB_ERR_NOMEM:
        raise MemoryError(f'libcyten_amd: {msg}')
    if status == CYB_ERR_UNSUPPORTED:
        raise NotImplementedError(f'libcyten_amd: {msg}')
    raise CybError(status, msg)
