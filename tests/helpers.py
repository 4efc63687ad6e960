"""Shared helpers of the test-suite: plain-data tensors <-> device tensors, invariant checks."""
import numpy as np

from cyten_amd import abelian as ab
from cyten_amd import workloads as wl


def to_device_tensor(bb, spec: wl.TensorSpec) -> ab.AbelianTensor:
    return ab.AbelianTensor.from_spec(bb, spec)


def check_svd_invariants(a, U, S, Vh, tol=1e-10, sref=None):
    """The reference's SVD acceptance criteria (tests/python_tests/test_tensors.py:3405-3500):
    S >= 0 and descending, |S| = |A|, U S Vh = A, U^T U = 1, Vh Vh^T = 1 -- plus, with `sref`
    (LAPACK singular values), |S - sref| <= tol * |A|."""
    k = min(a.shape)
    assert U.shape == (a.shape[0], k) and S.shape == (k,) and Vh.shape == (k, a.shape[1])
    nrm = max(np.linalg.norm(a), 1e-300)
    assert np.all(S >= 0)
    assert np.all(S[:-1] >= S[1:] - tol * nrm)
    assert abs(np.linalg.norm(S) - np.linalg.norm(a)) <= tol * nrm
    assert np.abs((U * S) @ Vh - a).max() <= tol * nrm
    assert np.abs(U.T @ U - np.eye(k)).max() <= tol
    assert np.abs(Vh @ Vh.T - np.eye(k)).max() <= tol
    if sref is not None:
        assert np.abs(S - sref).max() <= tol * nrm
