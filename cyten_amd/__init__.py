"""cyten_amd -- MI355X-native block backend for the charge-block-sparse tdot / SVD / QR / eigh
hot path of cyten.

Layout: ``csrc/`` hand-written HIP for gfx950 behind the C-ABI of ``include/cyten_amd.h``;
``block_backend`` the host-side mirror of cyten's ``BlockBackend`` operator API;
``deferred`` the same backend with lazy per-block ``matrix_dot`` / decompositions (the reference's unchanged
call sites); ``abelian`` the sector bookkeeping callers; ``krylov`` the Lanczos matvec group; ``sharding``
sector sharding over the GPUs of a node;
``workloads`` the synthetic BASELINE inputs.  Importing the package does not load the HIP
library; constructing a :class:`HipBlockBackend` does, and fails loudly if it is missing.
"""
__version__ = '0.1.0'

from . import abelian, krylov, sharding, workloads  # noqa: F401
from .block_backend import GemmPlan, HipBlock, HipBlockBackend, Scalar  # noqa: F401
