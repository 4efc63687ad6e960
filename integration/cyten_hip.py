"""Registration of the HIP block backend with an unmodified cyten (SURVEY.md 8b "Registration").

    import integration.cyten_hip as hip
    hip.register()                                   # once per process, before cyten.get_backend(..., 'hip')
    be = cyten.get_backend('abelian', 'hip')         # AbelianBackend / FusionTreeBackend / NoSymmetryBackend on the device

How it plugs in (all file:line under /root/reference):
* the block backend is a Python subclass of ``cyten._core.ArrayApiBlockBackend`` (include/cyten/block_backend/array_api.h:12-16;
  trampoline ``PyArrayApiBlockBackend`` pybind/block_backend/py_trampolines.hpp:496-...) constructed over
  :class:`integration.hip_array_api.HipArrayNamespace`; the C++ base class turns every ``BlockBackend`` virtual into calls
  on that namespace, so blocks are its own ``ArrayApiBlockBackend.BlockCls`` holding HipArray handles (the abstract
  ``BlockBackend.BlockCls`` has no constructor bound for Python -- pybind/block_backend/py_block_backend.cpp:81-83 --, so a
  pure-Python Block subclass is not an option; this route needs no such class);
* the subclass adds the eight operations the reference base leaves ``NotImplemented`` (array_api.cpp:678,782,788,838,934,
  1002,1111 and the N-d ``tile``) -- none of them is on the tdot / SVD / QR / eigh path.  The reference's own example for
  such overrides is ``to_numpy`` in, ``block_from_numpy`` out (tests/python_tests/backends/test_array_api_block_backend.py:
  33-43); since round 3 they run on the device kernels instead and only fall back to that pattern for host-held arrays.  Override names equal the method names (the trampoline looks the Python attribute up by the
  C++ method name: SURVEY.md section 7, hard part 7);
* ``get_backend(symmetry, 'hip')`` works without touching cyten because ``backend_factory.cpp:58-67,100-104`` consults
  the Python dict ``cyten._core._tensor_backend_cache`` keyed ``(tensor_backend_str, block_backend_str)`` first; foreign
  backends are held by a no-op deleter (pybind/backends/py_abelian.cpp:33-35), so the objects are kept alive HERE.

Nothing in this module is imported by the product: cyten is not installed where the product is built and benchmarked.
"""
from __future__ import annotations

import numpy as np

_alive = {}          # device -> (namespace, block backend, {tensor backend name: instance}): process lifetime

#: the operations ``ArrayApiBlockBackend`` leaves to a Python subclass, with the numpy routine NumpyBlockBackend uses
COLD_OVERRIDES = ('angle', 'block_from_diagonal', 'block_from_mask', 'kron', 'real_if_close', 'sqrt', 'matrix_exp', 'tile')


def _make_backend_class(core):
    """Built lazily: the base class only exists once cyten is imported."""
    import scipy.linalg

    from .hip_array_api import HipArray

    class HipArrayApiBlockBackend(core.ArrayApiBlockBackend):
        """cyten block backend on libcyten_amd (MI355X).  See the module docstring."""

        def __init__(self, namespace, default_device='cuda:0'):
            core.ArrayApiBlockBackend.__init__(self, namespace, default_device)
            self._xp = namespace

        def get_backend_name(self):
            return 'HipArrayApiBlockBackend'

        def synchronize_device(self):
            """cyten's ``synchronize()`` is an empty non-virtual (block_backend.cpp:1042-1045); callers that time kernels
            call this instead."""
            self._xp.bb.synchronize()

        # -- the operations the C++ base leaves open.  Round 3: they run on the DEVICE kernels of HipBlockBackend; a block's
        #    array is reached through Block::to_numpy under the namespace's `passthrough` (no host copy), the result goes back
        #    through as_block (array_api.cpp:568-600: api.asarray of an array object is the identity).  Host-held arrays
        #    (index data: int64) take the reference's own pattern, numpy in / block_from_numpy out.
        def _dev(self, blk):
            with self._xp.passthrough():
                got = self._xp.unbox(blk.to_numpy())
            return got if isinstance(got, HipArray) else self._xp.asarray(got)

        def _out(self, hip_block):
            return self.as_block(HipArray(self._xp, hip_block))

        def _unary(self, a, dev_fn, host_fn):
            x = self._dev(a)
            if x.blk is None:
                return self.block_from_numpy(host_fn(x.host))
            return self._out(dev_fn(x.blk))

        def angle(self, a):
            return self._unary(a, self._xp.bb.angle, np.angle)

        def sqrt(self, a):
            return self._unary(a, self._xp.bb.sqrt, np.sqrt)

        def block_from_diagonal(self, diag):
            return self._unary(diag, self._xp.bb.block_from_diagonal, np.diag)

        def matrix_exp(self, matrix):
            return self._unary(matrix, self._xp.bb.matrix_exp, scipy.linalg.expm)

        def real_if_close(self, a, tol):
            return self._unary(a, lambda x: self._xp.bb.real_if_close(x, tol), lambda x: np.real_if_close(x, tol=tol))

        def kron(self, a, b):
            x, y = self._dev(a), self._dev(b)
            if x.blk is None or y.blk is None:
                return self.block_from_numpy(np.kron(np.asarray(x), np.asarray(y)))
            return self._out(self._xp.bb.kron(x.blk, y.blk))

        def tile(self, a, repeats):
            x = self._dev(a)
            if x.blk is None or x.ndim != 1 or not isinstance(repeats, (int, np.integer)):
                return self.block_from_numpy(np.tile(np.asarray(x), repeats))
            return self._out(self._xp.bb.tile(x.blk, int(repeats)))

        def block_from_mask(self, mask, dtype):
            m = self._dev(mask)                          # (a length-M boolean vector: read once, it sizes the result)
            flags = np.asarray(m, dtype=bool)
            return self._out(self._xp.bb.block_from_mask(flags, dtype.to_numpy_dtype()))

    return HipArrayApiBlockBackend


def register(name: str = 'hip', device: str = 'cuda:0', deferred: bool = True):
    """Create the block backend for `device` and seed cyten's backend cache so that ``cyten.get_backend(sym, name)``
    returns tensor backends built on it.  Returns the block backend.  Idempotent per device."""
    import cyten                                     # noqa: F401  (lazy: only where cyten exists)
    from cyten import _core as core

    from .hip_array_api import HipArrayNamespace
    if device in _alive:
        return _alive[device][1]
    xp = HipArrayNamespace(device, deferred=deferred)
    bb = _make_backend_class(core)(xp, xp.device)
    tensor_backends = {'abelian': core.AbelianBackend(bb), 'fusion_tree': core.FusionTreeBackend(bb),
                       'no_symmetry': core.NoSymmetryBackend(bb)}
    for tb_name, tb in tensor_backends.items():
        core._tensor_backend_cache[(tb_name, name)] = tb       # backend_factory.cpp:58-67 looks here first
    _alive[device] = (xp, bb, tensor_backends)                 # no-op deleter on the C++ side: we own the lifetime
    return bb


CONFTEST_SNIPPET = '''
# conftest.py for parity runs of cyten's own test-suite with this backend (reference: /root/reference/conftest.py:160-162
# option --block-backends, :226-238 the table of legal names): put this file next to cyten's conftest.py, or paste it in.
import pytest
import conftest as cyten_conftest                      # cyten's conftest module

cyten_conftest._block_backend_params['hip'] = pytest.param('hip')      # `--block-backends hip` becomes a legal choice


def pytest_sessionstart(session):
    import integration.cyten_hip as hip
    hip.register('hip', 'cuda:0')                      # seeds cyten._core._tensor_backend_cache[(tensor backend, 'hip')]
'''
