"""TEST INFRASTRUCTURE (see oracle/__init__.py): CPU restatement of the reference's Lanczos ground-state
solver and of the two-site effective Hamiltonian, on dense numpy arrays.

* ``lanczos_dense`` follows /root/reference/src/tensors/krylov_based.cpp:803-946 (LanczosGroundState:
  _build_krylov :843-883, _converged :885-894, _calc_result_krylov :922-946, _calc_result_full :310-341)
  with the option defaults of :276-288.  Pinned in tests/test_oracle.py against numpy.linalg.eigh on
  seeded symmetric matrices (the reference's own test does the same: tests/python_tests/test_krylov_based.py
  compares with the dense spectrum).
* ``heff_dense`` is the contraction of toycodes/tenpy_toycodes/d_dmrg.py:74-86 written as one einsum over
  the dense versions of LP, W1, W2, RP (leg orders of cyten_amd.krylov).
"""
import numpy as np


def heff_dense(LP, W1, W2, RP):
    """Returns matvec(theta) for dense arrays LP[x,l,y], W1[i,c,j,l], W2[k,d,m,c], RP[d,z,w],
    theta[y,j,m,z] -> theta'[x,i,k,w]."""
    def matvec(theta):
        # the order of d_dmrg.py:74-86 (numpy's path optimiser picks an O(chi^4) route for the one-shot einsum)
        t = np.einsum('xly,yjmz->xljmz', LP, theta)
        t = np.einsum('icjl,xljmz->xicmz', W1, t)
        t = np.einsum('kdmc,xicmz->xikdz', W2, t)
        return np.einsum('dzw,xikdz->xikw', RP, t)
    return matvec


def heff_matrix(LP, W1, W2, RP):
    """The dense H_eff matrix over the flattened (x,i,k,w) index (small sizes only)."""
    H = np.einsum('xly,icjl,kdmc,dzw->xikwyjmz', LP, W1, W2, RP, optimize=True)
    n = H.shape[0] * H.shape[1] * H.shape[2] * H.shape[3]
    return H.reshape(n, n)


def lanczos_dense(matvec, psi0, N_min=2, N_max=20, P_tol=1e-14, min_gap=1e-12, reortho=False, cutoff=None,
                  E_tol=np.inf):
    """(E0, psi, N) for a Hermitian `matvec` acting on numpy arrays of psi0's shape."""
    if cutoff is None:
        cutoff = np.finfo(np.float64).eps * 100
    h = np.zeros((N_max + 1, N_max + 1))
    Es = np.zeros((N_max, N_max))
    cache = []
    w = psi0
    beta = np.linalg.norm(w)
    if beta < cutoff:
        raise ValueError(f'Norm of self.psi0 too small: {beta}')
    psi0 = w / beta
    vf = np.ones(1)
    N = 0
    for k in range(N_max):
        w = w / beta
        cache.append(w)
        w = matvec(w)
        alpha = float(np.real(np.vdot(cache[-1], w)))   # krylov_based.cpp:861: inner(w, cache.back()).real()
        h[k, k] = alpha
        if k == 0:
            Es[0, 0] = alpha
            vf = np.ones(1)
        else:
            E_kr, v_kr = np.linalg.eigh(h[:k + 1, :k + 1])
            Es[k, :k + 1] = E_kr
            vf = v_kr[:, 0].copy()
        w = w - alpha * cache[-1]
        if reortho:
            for v in cache[:-1]:
                w = w - np.vdot(v, w) * v
        elif k > 0:
            w = w - beta * cache[-2]
        beta = np.linalg.norm(w)
        h[k, k + 1] = h[k + 1, k] = beta
        N = k + 1
        if abs(beta) < cutoff:
            break
        if k + 1 >= N_min:
            ritz = abs(vf[k]) * abs(h[k, k + 1])
            gap = max(Es[k, 1] - Es[k, 0], min_gap)
            if (ritz / gap) ** 2 < P_tol and Es[k - 1, 0] - Es[k, 0] < E_tol:
                break
    E0 = Es[N - 1, 0]
    if N == 1:
        return E0, psi0, N
    psif = vf[0] * psi0
    for k in range(1, N):
        psif = psif + vf[N - k] * cache[len(cache) - k]
    return E0, psif / np.linalg.norm(psif), N
