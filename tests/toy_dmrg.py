"""Two-site DMRG of the transverse-field Ising chain on the block-sparse path of this repo (TEST INFRASTRUCTURE).

The reference pins its whole abelian stack with a known answer: ``test_dmrg_tfi``
(/root/reference/tests/python_tests/test_toycodes.py:92-105) runs its toy DMRG
(toycodes/tenpy_toycodes/d_dmrg.py) with Z2 conservation on an open TFI chain and compares the energy with exact
diagonalisation (b_model.py:175-206) to 1e-9.  This module is the same algorithm written against the interface of
``cyten_amd.abelian`` / ``cyten_amd.krylov`` -- compose, permute_legs, the Lanczos ground state of the two-site
effective Hamiltonian, combine_legs + batched SVD + truncation + split_legs, environment updates -- so the same known
answer pins the device path end to end (tests/test_toy_dmrg.py: numpy stand-in on the CPU, HIP backend on the GPU).

H = -J sum_i X_i X_{i+1} - g sum_i Z_i, Z2 charge = parity of the number of down spins.
Leg conventions (those of cyten_amd.krylov / workloads.config_heff; + incoming, - outgoing):
    A [vL+, p+, vR-]     W [p'+, wR+, p-, wL-]     LP [vL'+, wL+, vL-]     RP [wR-, vR+, vR'-]
MPO bond states: start (charge 0), done (charge 0), X-placed (charge 1); dense bond index start = 0, done = 1, X = 2.
"""
import numpy as np

from cyten_amd import abelian as ab
from cyten_amd import krylov

SYM = ab.Symmetry((2,))


def _leg(sectors, mults, sign):
    return ab.Leg(SYM, np.asarray(sectors, dtype=np.int64).reshape(len(mults), 1), mults, sign)


P_LEG = _leg([0, 1], [1, 1], +1)
W_LEG = _leg([0, 1], [2, 1], +1)
V0 = _leg([0], [1], +1)


def dense_to_tensor(bb, legs, dense, num_codomain=0):
    """Cut a dense array into its charge-allowed blocks (all-zero blocks are dropped)."""
    inds = ab.AbelianTensor.allowed_block_inds(SYM, legs)
    blocks, rows = [], []
    for row in inds:
        sl = tuple(slice(int(l.slices[i]), int(l.slices[i + 1])) for l, i in zip(legs, row))
        blk = np.ascontiguousarray(dense[sl])
        if np.any(blk != 0.0):
            blocks.append(blk)
            rows.append(row)
    # everything outside the allowed blocks must vanish: the operator respects the symmetry
    mask = np.zeros(dense.shape, dtype=bool)
    for row in inds:
        mask[tuple(slice(int(l.slices[i]), int(l.slices[i + 1])) for l, i in zip(legs, row))] = True
    assert not np.any(dense[~mask] != 0.0), 'dense tensor violates the charge rule'
    return ab.AbelianTensor.from_numpy_blocks(bb, SYM, legs, blocks, np.array(rows, dtype=np.int64).reshape(len(rows), len(legs)),
                                              num_codomain)


def conj_tensor(bb, t):
    """Complex conjugate with dual legs (the bra tensor)."""
    blocks = [bb.conj(b) if hasattr(bb, 'conj') else np.conj(b) for b in t.blocks]
    return ab.AbelianTensor(t.symmetry, [l.dual() for l in t.legs], blocks, t.block_inds, t.num_codomain)


def tfi_mpo(bb, J, g):
    X = np.array([[0.0, 1.0], [1.0, 0.0]])
    Z = np.diag([1.0, -1.0])
    eye = np.eye(2)
    op = {(0, 0): eye, (0, 2): X, (0, 1): -g * Z, (2, 1): -J * X, (1, 1): eye}  # (wL, wR) in dense bond order
    W = np.zeros((2, 3, 2, 3))  # [p', wR, p, wL]
    for (wl, wr), o in op.items():
        W[:, wr, :, wl] = o
    return dense_to_tensor(bb, [P_LEG, W_LEG, P_LEG.dual(), W_LEG.dual()], W, 2)


def boundaries(bb):
    LP = np.zeros((1, 3, 1))
    LP[0, 0, 0] = 1.0  # bond state "start"
    RP = np.zeros((3, 1, 1))
    RP[1, 0, 0] = 1.0  # bond state "done"
    return (dense_to_tensor(bb, [V0, W_LEG, V0.dual()], LP, 2),
            dense_to_tensor(bb, [W_LEG.dual(), V0, V0.dual()], RP, 2))


def product_state(bb, L):
    """All spins up: bond dimension 1, charge 0 everywhere."""
    A = np.zeros((1, 2, 1))
    A[0, 0, 0] = 1.0
    return [dense_to_tensor(bb, [V0, P_LEG, V0.dual()], A, 2) for _ in range(L)]


def update_LP(bb, LP, A, W):
    """LP'[vC', wC, vC] = sum LP[vL', wL, vL] A[vL, p, vC] W[p', wC, p, wL] conj(A)[vL', p', vC']."""
    x = ab.compose(bb, LP, A, 1)                                  # [vL', wL, p, vC]
    x = ab.permute_legs(bb, x, [1, 2, 3, 0])                      # [wL, p, vC, vL']
    x = ab.compose(bb, W, x, 2)                                   # [p', wC, vC, vL']
    x = ab.permute_legs(bb, x, [1, 2, 0, 3])                      # [wC, vC, p', vL']
    x = ab.compose(bb, x, conj_tensor(bb, A), 2)                  # [wC, vC, vC']
    return ab.permute_legs(bb, x, [2, 0, 1], 2)                   # [vC', wC, vC]


def update_RP(bb, RP, B, W):
    """RP'[wC, vC, vC'] = sum B[vC, p, vR] RP[wR, vR, vR'] W[p', wR, p, wC] conj(B)[vC', p', vR']."""
    x = ab.compose(bb, B, ab.permute_legs(bb, RP, [1, 0, 2]), 1)  # [vC, p, wR, vR']
    x = ab.permute_legs(bb, x, [0, 3, 2, 1])                      # [vC, vR', wR, p]
    x = ab.compose(bb, x, ab.permute_legs(bb, W, [2, 1, 0, 3]), 2)  # W -> [p, wR, p', wC]: result [vC, vR', p', wC]
    x = ab.permute_legs(bb, x, [3, 0, 1, 2])                      # [wC, vC, vR', p']
    Bc = ab.permute_legs(bb, conj_tensor(bb, B), [1, 2, 0])       # [p', vR', vC']
    return ab.compose(bb, x, Bc, 2)                               # [wC, vC, vC']


def split_theta(bb, theta, chi_max, svd_min, absorb):
    """theta[vL, p0, p1, vR] -> A[vL, p0, vC], B[vC, p1, vR] by the truncated SVD of the path under test; the singular
    values (normalised) go into the right (`absorb='right'`) or left factor."""
    mv, U, S, Vh, err, _ = ab.truncated_svd(bb, theta, 2, chi_max=chi_max, svd_min=svd_min)
    s_np = [np.asarray(bb.to_numpy(s)) for s in S]
    nrm = np.sqrt(sum(float(np.sum(s ** 2)) for s in s_np))
    keep = [k for k, s in enumerate(s_np) if len(s)]
    charges = np.array([mv.charges[k] for k in keep], dtype=np.int64).reshape(len(keep), SYM.n)
    mults = [len(s_np[k]) for k in keep]
    vc_out = ab.Leg(SYM, charges, mults, -1)
    vc_in = vc_out.dual()
    pos = {tuple(int(x) for x in q): i for i, q in enumerate(vc_out.sectors)}
    new_index = {k: pos[tuple(int(x) for x in mv.charges[k])] for k in keep}
    Us, Vs = list(U), list(Vh)
    for k in keep:  # normalised singular values into the factor that carries the orthogonality centre
        s_blk = bb.mul(1.0 / nrm, S[k])
        if absorb == 'left':
            Us[k] = bb.scale_axis(U[k], s_blk, 1)
        else:
            Vs[k] = bb.scale_axis(Vh[k], s_blk, 0)
    a_blocks, a_rows, b_blocks, b_rows = [], [], [], []
    for sec, idx, blk in ab.split_matrix_legs(bb, mv, Us, 'rows'):
        if sec in new_index:
            a_blocks.append(blk)
            a_rows.append(list(idx) + [new_index[sec]])
    for sec, idx, blk in ab.split_matrix_legs(bb, mv, Vs, 'cols'):
        if sec in new_index:
            b_blocks.append(blk)
            b_rows.append([new_index[sec]] + list(idx))
    A = ab.AbelianTensor(SYM, [theta.legs[0], theta.legs[1], vc_out], a_blocks, np.array(a_rows, dtype=np.int64), 2).sorted()
    B = ab.AbelianTensor(SYM, [vc_in, theta.legs[2], theta.legs[3]], b_blocks, np.array(b_rows, dtype=np.int64), 2).sorted()
    return A, B, err


def dmrg(bb, L, J, g, chi_max=32, svd_min=1e-12, n_sweeps=6, lanczos_options=None):
    """Ground-state energy of the open TFI chain by two-site DMRG (d_dmrg.py:120-262: sweep right, sweep left)."""
    W = tfi_mpo(bb, J, g)
    psi = product_state(bb, L)
    LP0, RP0 = boundaries(bb)
    LPs, RPs = [None] * L, [None] * L
    LPs[0], RPs[L - 1] = LP0, RP0
    for i in range(L - 1, 1, -1):
        RPs[i - 1] = update_RP(bb, RPs[i], psi[i], W)
    opts = dict(N_max=30, P_tol=1e-14)
    opts.update(lanczos_options or {})
    energy = None
    for _ in range(n_sweeps):
        # right-moving half: the left factor is an isometry, the centre moves right; then back
        for i, right in [(i, True) for i in range(L - 1)] + [(i, False) for i in range(L - 2, -1, -1)]:
            theta = ab.compose(bb, psi[i], psi[i + 1], 1)
            H = krylov.HEffective(bb, LPs[i], W, W, RPs[i + 1])
            energy, theta, _ = krylov.lanczos(bb, H, theta, opts)
            psi[i], psi[i + 1], _ = split_theta(bb, theta, chi_max, svd_min, 'right' if right else 'left')
            if right:
                LPs[i + 1] = update_LP(bb, LPs[i], psi[i], W)
            else:
                RPs[i] = update_RP(bb, RPs[i + 1], psi[i + 1], W)
    return energy, psi


def tfi_exact_energy(L, J, g):
    """Exact diagonalisation of the same Hamiltonian (b_model.py:175-206 uses scipy's sparse eigsh likewise)."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import eigsh
    sx = sp.csr_matrix(np.array([[0.0, 1.0], [1.0, 0.0]]))
    sz = sp.csr_matrix(np.array([[1.0, 0.0], [0.0, -1.0]]))
    eye = sp.identity(2, format='csr')

    def site_op(o, i):
        out = sp.identity(1, format='csr')
        for k in range(L):
            out = sp.kron(out, o if k == i else eye, 'csr')
        return out
    H = sp.csr_matrix((2 ** L, 2 ** L))
    for i in range(L - 1):
        H = H - J * (site_op(sx, i) @ site_op(sx, i + 1))
    for i in range(L):
        H = H - g * site_op(sz, i)
    return float(eigsh(H, k=1, which='SA', return_eigenvectors=False, ncv=24)[0])
