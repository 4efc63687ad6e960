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
  1002,1111 and the N-d ``tile``) -- none of them is on the tdot / SVD / QR / eigh path; they follow the reference's own
  example for Python overrides (tests/python_tests/backends/test_array_api_block_backend.py:33-43: ``to_numpy`` in,
  ``block_from_numpy`` out).  Override names equal the method names (the trampoline looks the Python attribute up by the
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

        # -- the cold operations: numpy in, block out (the reference's own pattern for Python overrides)
        def angle(self, a):
            return self.block_from_numpy(np.angle(a.to_numpy()))

        def block_from_diagonal(self, diag):
            return self.block_from_numpy(np.diag(diag.to_numpy()))

        def block_from_mask(self, mask, dtype):
            m = np.asarray(mask.to_numpy(), dtype=bool)
            out = np.zeros((len(m), int(m.sum())), dtype=dtype.to_numpy_dtype())
            out[m, np.arange(int(m.sum()))] = 1
            return self.block_from_numpy(out)

        def kron(self, a, b):
            return self.block_from_numpy(np.kron(a.to_numpy(), b.to_numpy()))

        def real_if_close(self, a, tol):
            return self.block_from_numpy(np.real_if_close(a.to_numpy(), tol=tol))

        def sqrt(self, a):
            return self.block_from_numpy(np.sqrt(a.to_numpy()))

        def matrix_exp(self, matrix):
            return self.block_from_numpy(scipy.linalg.expm(matrix.to_numpy()))

        def tile(self, a, repeats):
            return self.block_from_numpy(np.tile(a.to_numpy(), repeats))

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
