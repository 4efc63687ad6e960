"""Two-site DMRG of the transverse-field Ising chain on the block-sparse path of this repo (TEST INFRASTRUCTURE).

The reference pins its whole abelian stack with a known answer: ``test_dmrg_tfi``
(/root/reference/tests/python_tests/test_toycodes.py:92-105) runs its toy DMRG
(toycodes/tenpy_toycodes/d_dmrg.py) with Z2 conservation on an open TFI chain and compares the energy with exact
diagonalisation (b_model.py:175-206) to 1e-9.  This module is the same algorithm written against the interface of
``cyten_amd.abelian`` / ``cyten_amd.krylov`` -- compose, permute_legs, the Lanczos ground state of the two-site
effective Hamiltonian, combine_legs + batched SVD + truncation + split_legs, environment updates -- so the same known
answer pins the device path end to end (tests/test_toy_dmrg.py: numpy stand-in on the CPU, HIP backend on the GPU).

Models: the TFI chain H = -J sum X X - g sum Z with the Z2 parity of the down spins, and the Heisenberg chain
H = J sum (X X + Y Y + Z Z) with U(1) = 2 Sz (test_dmrg_heisenberg of the reference, :58-71, has the same known answer).
Leg conventions (those of cyten_amd.krylov / workloads.config_heff; + incoming, - outgoing):
    A [vL+, p+, vR-]     W [p'+, wR+, p-, wL-]     LP [vL'+, wL+, vL-]     RP [wR-, vR+, vR'-]
"""
import numpy as np

from cyten_amd import abelian as ab
from cyten_amd import krylov

def _leg(sym, sectors, mults, sign):
    return ab.Leg(sym, np.asarray(sectors, dtype=np.int64).reshape(len(mults), 1), mults, sign)


class Model:
    """Symmetry, physical leg, MPO bond leg, dense MPO tensor W[p', wR, p, wL] (dense bond indices in the sorted sector
    order of the bond leg), the dense bond indices of the `start` and `done` states, and the initial product state."""

    def __init__(self, sym, p_leg, w_leg, W, start, done, init_state):
        self.sym, self.p_leg, self.w_leg, self.W, self.start, self.done, self.init_state = sym, p_leg, w_leg, W, start, done, init_state


def tfi_model(L, J, g):
    """H = -J sum X X - g sum Z with Z2 = parity of the down spins; bond states start (0), done (0), X placed (1)."""
    sym = ab.Symmetry((2,))
    X = np.array([[0.0, 1.0], [1.0, 0.0]])
    Z = np.diag([1.0, -1.0])
    eye = np.eye(2)
    op = {(0, 0): eye, (0, 2): X, (0, 1): -g * Z, (2, 1): -J * X, (1, 1): eye}  # (wL, wR): start = 0, done = 1, X = 2
    W = np.zeros((2, 3, 2, 3))
    for (wl, wr), o in op.items():
        W[:, wr, :, wl] = o
    return Model(sym, _leg(sym, [0, 1], [1, 1], +1), _leg(sym, [0, 1], [2, 1], +1), W, 0, 1, [0] * L)


def heisenberg_model(L, J):
    """H = J sum (X X + Y Y + Z Z) = J sum (2 s+ s- + 2 s- s+ + Z Z) with U(1) = 2 Sz (b_model.py:209-246 diagonalises the
    same operator).  Physical charges: up +1, down -1 (sorted: down first).  Bond states and charges (an operator that
    raises the charge by dq leads to a bond state of charge -dq): P = after s+ (-2), start / done / Z (0), M = after s- (+2);
    dense bond order P = 0, start = 1, done = 2, Z = 3, M = 4.  Initial state: Neel."""
    sym = ab.Symmetry((0,))
    p_leg = _leg(sym, [-1, 1], [1, 1], +1)          # dense physical index 0 = down, 1 = up
    w_leg = _leg(sym, [-2, 0, 2], [1, 3, 1], +1)
    sp = np.array([[0.0, 0.0], [1.0, 0.0]])          # |up><down| in the (down, up) basis
    sm = sp.T
    Z = np.diag([-1.0, 1.0])
    eye = np.eye(2)
    P, START, DONE, ZS, M = 0, 1, 2, 3, 4
    op = {(START, START): eye, (DONE, DONE): eye, (START, P): sp, (P, DONE): 2.0 * J * sm, (START, M): sm, (M, DONE): 2.0 * J * sp,
          (START, ZS): Z, (ZS, DONE): J * Z}
    W = np.zeros((2, 5, 2, 5))
    for (wl, wr), o in op.items():
        W[:, wr, :, wl] = o
    return Model(sym, p_leg, w_leg, W, START, DONE, [1 if i % 2 == 0 else 0 for i in range(L)])


def dense_to_tensor(bb, sym, legs, dense, num_codomain=0):
    """Cut a dense array into its charge-allowed blocks (all-zero blocks are dropped)."""
    inds = ab.AbelianTensor.allowed_block_inds(sym, legs)
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
    return ab.AbelianTensor.from_numpy_blocks(bb, sym, legs, blocks, np.array(rows, dtype=np.int64).reshape(len(rows), len(legs)),
                                              num_codomain)


def conj_tensor(bb, t):
    """Complex conjugate with dual legs (the bra tensor)."""
    blocks = [bb.conj(b) if hasattr(bb, 'conj') else np.conj(b) for b in t.blocks]
    return ab.AbelianTensor(t.symmetry, [l.dual() for l in t.legs], blocks, t.block_inds, t.num_codomain)


def mpo_tensor(bb, model):
    return dense_to_tensor(bb, model.sym, [model.p_leg, model.w_leg, model.p_leg.dual(), model.w_leg.dual()], model.W, 2)


def product_state(bb, model):
    """Product state with one basis state per site; the bond legs carry the accumulated charge (dimension 1)."""
    sym, p = model.sym, model.p_leg
    tensors, q = [], np.zeros(sym.n, dtype=np.int64)
    for k in model.init_state:
        sec = int(np.searchsorted(p.slices, k, side='right') - 1)   # sector of dense physical index k
        vl = _leg(sym, [q.copy()], [1], +1)
        q = sym.reduce(q + p.sign * p.sectors[sec])
        vr = _leg(sym, [q.copy()], [1], -1)
        A = np.zeros((1, p.dim, 1))
        A[0, k, 0] = 1.0
        tensors.append(dense_to_tensor(bb, sym, [vl, p, vr], A, 2))
    return tensors


def boundaries(bb, model, psi):
    sym, w = model.sym, model.w_leg
    vl, vr = psi[0].legs[0], psi[-1].legs[2]
    LP = np.zeros((1, w.dim, 1))
    LP[0, model.start, 0] = 1.0
    RP = np.zeros((w.dim, 1, 1))
    RP[model.done, 0, 0] = 1.0
    return (dense_to_tensor(bb, sym, [vl, w, vl.dual()], LP, 2),
            dense_to_tensor(bb, sym, [w.dual(), vr.dual(), vr], RP, 2))


def _recorded(bb, cache, tag, fn, tensors):
    """``fn()`` through the launch-recording cache of the HIP backend (cyten_amd/replay.py); plain call elsewhere."""
    if cache is None or not hasattr(bb, 'ctx'):
        return fn()
    from cyten_amd.replay import apply_recorded
    return apply_recorded(bb, cache, tag, fn, tensors)[0]


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
    sym = theta.symmetry
    charges = np.array([mv.charges[k] for k in keep], dtype=np.int64).reshape(len(keep), sym.n)
    mults = [len(s_np[k]) for k in keep]
    vc_out = ab.Leg(sym, charges, mults, -1)
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
    A = ab.AbelianTensor(sym, [theta.legs[0], theta.legs[1], vc_out], a_blocks, np.array(a_rows, dtype=np.int64), 2).sorted()
    B = ab.AbelianTensor(sym, [vc_in, theta.legs[2], theta.legs[3]], b_blocks, np.array(b_rows, dtype=np.int64), 2).sorted()
    return A, B, err


def dmrg(bb, model, chi_max=32, svd_min=1e-12, n_sweeps=6, lanczos_options=None, sweep_times=False, stats=None, on_sweep=None):
    """Ground-state energy of an open chain by two-site DMRG (d_dmrg.py:120-262: sweep right, sweep left)."""
    W = mpo_tensor(bb, model)
    psi = product_state(bb, model)
    L = len(psi)
    LP0, RP0 = boundaries(bb, model, psi)
    LPs, RPs = [None] * L, [None] * L
    LPs[0], RPs[L - 1] = LP0, RP0
    for i in range(L - 1, 1, -1):
        RPs[i - 1] = update_RP(bb, RPs[i], psi[i], W)
    opts = dict(N_max=30, P_tol=1e-14)
    opts.update(lanczos_options or {})
    energy = None
    import time
    times = []
    heff_cache = {}   # recorded H_eff launch sequences, shared by all bonds and sweeps (krylov.HEffective)
    for sweep in range(n_sweeps):
        if on_sweep is not None:
            on_sweep(sweep)   # (profiling hook of scripts/dmrg_profile.py)
        t_sweep = time.perf_counter()
        # right-moving half: the left factor is an isometry, the centre moves right; then back
        for i, right in [(i, True) for i in range(L - 1)] + [(i, False) for i in range(L - 2, -1, -1)]:
            theta = _recorded(bb, heff_cache, 'theta', lambda: ab.compose(bb, psi[i], psi[i + 1], 1), [psi[i], psi[i + 1]])
            H = krylov.HEffective(bb, LPs[i], W, W, RPs[i + 1], cache=heff_cache)
            energy, theta, _ = krylov.lanczos(bb, H, theta, opts)
            if stats is not None:
                stats['recorded'] = stats.get('recorded', 0) + getattr(H, 'n_recorded', 0)
                stats['replayed'] = stats.get('replayed', 0) + getattr(H, 'n_replayed', 0)
            psi[i], psi[i + 1], _ = split_theta(bb, theta, chi_max, svd_min, 'right' if right else 'left')
            if right:
                LPs[i + 1] = _recorded(bb, heff_cache, 'LP', lambda: update_LP(bb, LPs[i], psi[i], W), [LPs[i], psi[i], W])
            else:
                RPs[i] = _recorded(bb, heff_cache, 'RP', lambda: update_RP(bb, RPs[i + 1], psi[i + 1], W), [RPs[i + 1], psi[i + 1], W])
        if hasattr(bb, 'synchronize'):
            bb.synchronize()
        times.append(time.perf_counter() - t_sweep)
    return (energy, psi, times) if sweep_times else (energy, psi)


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


def heisenberg_exact_energy(L, J):
    """Exact diagonalisation of J sum (X X + Y Y + Z Z) (b_model.py:209-246)."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import eigsh
    sx = sp.csr_matrix(np.array([[0.0, 1.0], [1.0, 0.0]]))
    sy = sp.csr_matrix(np.array([[0.0, -1.0j], [1.0j, 0.0]]))
    sz = sp.csr_matrix(np.array([[1.0, 0.0], [0.0, -1.0]]))
    eye = sp.identity(2, format='csr')

    def site_op(o, i):
        out = sp.identity(1, format='csr')
        for k in range(L):
            out = sp.kron(out, o if k == i else eye, 'csr')
        return out
    H = sp.csr_matrix((2 ** L, 2 ** L), dtype=complex)
    for i in range(L - 1):
        for o in (sx, sy, sz):
            H = H + J * (site_op(o, i) @ site_op(o, i + 1))
    return float(np.real(eigsh(H, k=1, which='SA', return_eigenvectors=False, ncv=24)[0]))
