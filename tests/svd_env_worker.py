"""Child process of tests/test_gpu_decomp.py::test_svd_pipeline_variants: the batched SVD of a list that goes through the
QR-preconditioned pipeline, under whatever CYB_SVD_* switches the parent put into the environment (they are read once per
process).  Prints OK or raises."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import check_svd_invariants   # noqa: E402
from cyten_amd.block_backend import HipBlockBackend   # noqa: E402

bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(7)
q1, _ = np.linalg.qr(rng.standard_normal((140, 140)))
q2, _ = np.linalg.qr(rng.standard_normal((140, 140)))
dep = rng.standard_normal((150, 100))
dep[:, :40] = dep[:, 40:80] @ rng.standard_normal((40, 40))
mats = [rng.standard_normal((300, 120)) @ rng.standard_normal((120, 260)),     # theta-like, rank-deficient
        (q1 * np.logspace(0, -14, 140)) @ q2,                                   # DMRG-like graded spectrum
        rng.standard_normal((700, 64)), rng.standard_normal((70, 500)), rng.standard_normal((200, 200)),
        dep, np.ones((96, 64)), np.zeros((96, 64)), np.eye(128), rng.standard_normal((5, 7)),
        np.outer(rng.standard_normal(90), rng.standard_normal(110)),
        rng.standard_normal((1700, 64)), rng.standard_normal((40, 1650))]                     # panels of more than 1536 rows
res = bb.matrix_svd_batched([bb.as_block(m) for m in mats])
for m, (U, S, Vh) in zip(mats, res):
    check_svd_invariants(m, bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh), 1e-10, sref=np.linalg.svd(m, compute_uv=False))

# complex128 blocks on the same pipeline through the interleaved embedding (CYB_SVD_EMBEDDED_COMPLEX): full rank, tall /
# wide rank deficiency (pair-wise deflation, completion), degenerate and graded spectra, one null direction
def crandn(shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


u1, _ = np.linalg.qr(crandn((120, 90)))
u2, _ = np.linalg.qr(crandn((90, 90)))
cm = [crandn((130, 130)), crandn((260, 70)), crandn((70, 260)), crandn((180, 40)) @ crandn((40, 150)), crandn((150, 40)) @ crandn((40, 180)),
      (u1 * np.repeat([3.0, 2.0, 1.0], 30)) @ u2.conj().T, (u1 * np.logspace(0, -14, 90)) @ u2.conj().T,
      crandn((90, 89)) @ crandn((89, 90)), crandn((900, 48))]
got = bb._complex_svd_embedded(bb.contiguous_many([bb.as_block(m) for m in cm]))
if got is None:   # the embedded route needs the persistent sweep kernel: without it the engine refuses, the caller falls back
    assert 'CYB_JACOBI_NOSWEEP' in os.environ
    got = (bb.matrix_svd_batched([bb.as_block(m) for m in cm]),)
for m, (U, S, Vh) in zip(cm, got[0]):
    U, S, Vh = bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh)
    k = min(m.shape)
    nrm = np.linalg.norm(m)
    assert U.dtype == np.complex128 and S.dtype == np.float64 and np.all(S[:-1] >= S[1:])
    assert np.abs((U * S) @ Vh - m).max() <= 1e-10 * nrm
    assert np.abs(U.conj().T @ U - np.eye(k)).max() <= 1e-10 and np.abs(Vh @ Vh.conj().T - np.eye(k)).max() <= 1e-10
    assert np.abs(S - np.linalg.svd(m, compute_uv=False)).max() <= 1e-10 * nrm
print('OK')
