"""Device runtime: one C-ABI context per GPU, device memory through torch's caching allocator.

PyTorch is plumbing here (device memory, the current HIP stream, torch.distributed); every
arithmetic or data-movement operation on block data goes through libcyten_amd's C-ABI.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib

_lock = threading.Lock()
_contexts: dict[int, 'Context'] = {}


class Context:
    """Per-device context (mirrors the per-device backend singletons of the reference,
    ``NumpyBlockBackend::from_factory`` numpy.cpp:411-432, ``TorchBlockBackend`` torch.cpp:669-695).
    """

    def __init__(self, device_index: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('cyten_amd needs a HIP device (torch.cuda.is_available() is False); '
                               'there is no CPU fallback')
        self.torch = torch
        self.lib = _lib.load()
        self.device_index = int(device_index)
        self.device = torch.device('cuda', self.device_index)
        torch.cuda.set_device(self.device)
        self._stream_ptr = torch.cuda.current_stream(self.device).cuda_stream
        # (the raw getter answers in ~0.2 us; `current_stream().cuda_stream` builds a Stream object per call: ~5 us,
        #  fifty times per DMRG bond update)
        self._raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        self.handle = C.c_void_p()
        _lib.check(self.lib.cyb_ctx_create(C.byref(self.handle), self.device_index, C.c_void_p(self._stream_ptr)))
        n_cu, lds = C.c_int(), C.c_int()
        hbm = C.c_int64()
        arch = C.create_string_buffer(64)
        _lib.check(self.lib.cyb_device_info(self.handle, C.byref(n_cu), C.byref(lds), C.byref(hbm), arch, 64))
        self.n_cu, self.lds_bytes, self.hbm_bytes, self.arch = n_cu.value, lds.value, hbm.value, arch.value.decode()

    # -- stream handling: always follow torch's current stream so torch events/timers see us
    def sync_stream(self):
        raw = self._raw_stream
        ptr = raw(self.device_index) if raw is not None else self.torch.cuda.current_stream(self.device).cuda_stream
        if ptr != self._stream_ptr:
            _lib.check(self.lib.cyb_ctx_set_stream(self.handle, C.c_void_p(ptr)))
            self._stream_ptr = ptr

    def synchronize(self):
        self.sync_stream()
        _lib.check(self.lib.cyb_ctx_sync(self.handle))

    # -- memory
    def empty(self, n: int, dtype='float64'):
        """Uninitialised device array of `n` elements (torch caching allocator)."""
        tdt = {'float64': self.torch.float64, 'complex128': self.torch.complex128, 'int64': self.torch.int64,
               'uint8': self.torch.uint8, 'int32': self.torch.int32, 'bool': self.torch.bool}[dtype]
        return self.torch.empty(max(int(n), 1), dtype=tdt, device=self.device)

    def h2d(self, dst_tensor, src: np.ndarray, dst_offset_elems: int = 0):
        src = np.ascontiguousarray(src)
        if src.nbytes == 0:
            return
        self.sync_stream()
        ptr = dst_tensor.data_ptr() + dst_offset_elems * dst_tensor.element_size()
        _lib.check(self.lib.cyb_memcpy_h2d(self.handle, C.c_void_p(ptr), src.ctypes.data_as(C.c_void_p), src.nbytes))

    def d2h(self, src_tensor, n_elems: int, np_dtype, src_offset_elems: int = 0) -> np.ndarray:
        out = np.empty(int(n_elems), dtype=np_dtype)
        if out.nbytes:
            self.sync_stream()
            ptr = src_tensor.data_ptr() + src_offset_elems * src_tensor.element_size()
            _lib.check(self.lib.cyb_memcpy_d2h(self.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        return out

    # -- events (HIP events on the context's stream)
    def event(self):
        ev = C.c_void_p()
        _lib.check(self.lib.cyb_event_create(C.byref(ev)))
        return ev

    def record(self, ev):
        self.sync_stream()
        _lib.check(self.lib.cyb_event_record(self.handle, ev))

    def time_next_gemm(self, e0, e1):
        """the next asynchronous grouped-GEMM launch records e0 / e1 directly around its kernel (behind the descriptor upload)"""
        _lib.check(self.lib.cyb_ctx_time_next_gemm(self.handle, e0, e1))

    def elapsed_ms(self, e0, e1) -> float:
        ms = C.c_float()
        _lib.check(self.lib.cyb_event_elapsed_ms(e0, e1, C.byref(ms)))
        return float(ms.value)

    def mfma_f64_peak(self, iters: int = 400000, waves_per_simd: int = 4, n_acc: int = 4):
        """Measured back-to-back v_mfma_f64_16x16x4_f64 rate of the chip (TFLOP/s, ms)."""
        tf, ms = C.c_double(), C.c_double()
        self.sync_stream()
        _lib.check(self.lib.cyb_mfma_f64_peak(self.handle, iters, n_acc * 100 + waves_per_simd, C.byref(tf), C.byref(ms)))
        return tf.value, ms.value


def get_context(device_index: int | None = None) -> Context:
    import torch
    if device_index is None:
        device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    with _lock:
        ctx = _contexts.get(device_index)
        if ctx is None:
            ctx = Context(device_index)
            _contexts[device_index] = ctx
        return ctx
