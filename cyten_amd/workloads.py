"""Synthetic inputs of the BASELINE.json configs, as plain numpy data.

Everything here is host data (sector tables, block index tables, numpy blocks seeded with
``numpy.random.default_rng``) so that the same inputs can be fed to the HIP path, to the oracle,
and to the CPU baseline.  No device code, no oracle import.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

DEFAULT_SEED = 12345  # the reference's test default (conftest.py:162)


@dataclass
class LegSpec:
    sectors: np.ndarray  # (nsec, nsym) int64
    mults: np.ndarray    # (nsec,) int64
    sign: int


@dataclass
class TensorSpec:
    """A block-sparse tensor as plain data."""
    moduli: tuple
    legs: list            # [LegSpec]
    block_inds: np.ndarray
    blocks: list          # numpy arrays
    num_codomain: int = 0


def _lexsort_rows(a):
    return np.lexsort(a.T) if a.shape[0] else np.zeros(0, dtype=np.int64)


def make_leg(moduli, sectors, mults, sign=+1) -> LegSpec:
    sectors = np.asarray(sectors, dtype=np.int64).reshape(len(mults), len(moduli))
    for k, m in enumerate(moduli):
        if m:
            sectors[:, k] %= m
    mults = np.asarray(mults, dtype=np.int64)
    order = _lexsort_rows(sectors)
    return LegSpec(sectors[order], mults[order], sign)


def u1_leg(chi: int, sigma_q: float = 2.0, sign=+1) -> LegSpec:
    """SURVEY 8d leg generator: charges q in [-4 sigma, 4 sigma], m_q = floor(chi w_q / sum w),
    w_q = exp(-q^2 / 2 sigma^2), remainder to q = 0, drop empty sectors."""
    qmax = int(np.floor(4 * sigma_q))
    qs = np.arange(-qmax, qmax + 1)
    w = np.exp(-qs ** 2 / (2.0 * sigma_q ** 2))
    m = np.floor(chi * w / w.sum()).astype(np.int64)
    m[qs == 0] += chi - m.sum()
    keep = m > 0
    return make_leg((0,), qs[keep][:, None], m[keep], sign)


def u1u1_leg(chi: int, sigma_n: float = 2.0, sigma_s: float = 1.5, sign=+1) -> LegSpec:
    """cfg3 leg: (n, s) with n + s even, Gaussian weights (charge + 2Sz)."""
    nmax, smax = int(np.floor(4 * sigma_n)), int(np.floor(4 * sigma_s))
    secs, ws = [], []
    for n in range(-nmax, nmax + 1):
        for s in range(-smax, smax + 1):
            if (n + s) % 2 == 0:
                secs.append((n, s))
                ws.append(np.exp(-n * n / (2 * sigma_n ** 2) - s * s / (2 * sigma_s ** 2)))
    secs, ws = np.array(secs), np.array(ws)
    m = np.floor(chi * ws / ws.sum()).astype(np.int64)
    m[np.flatnonzero(np.all(secs == 0, axis=1))[0]] += chi - m.sum()
    keep = m > 0
    return make_leg((0, 0), secs[keep], m[keep], sign)


def allowed_block_inds(moduli, legs) -> np.ndarray:
    grids = np.indices([len(l.mults) for l in legs]).reshape(len(legs), -1).T
    tot = np.zeros((grids.shape[0], len(moduli)), dtype=np.int64)
    for k, l in enumerate(legs):
        tot += l.sign * l.sectors[grids[:, k]]
    for k, m in enumerate(moduli):
        if m:
            tot[:, k] %= m
    inds = grids[np.all(tot == 0, axis=1)].astype(np.int64)
    return inds[_lexsort_rows(inds)]


def random_tensor(moduli, legs, rng, num_codomain=0, fill=1.0) -> TensorSpec:
    """All charge-allowed blocks, standard normal entries (a fraction `fill` of them present)."""
    inds = allowed_block_inds(moduli, legs)
    if fill < 1.0 and len(inds):
        keep = rng.random(len(inds)) < fill
        inds = inds[keep]
    blocks = [rng.standard_normal([int(l.mults[i]) for l, i in zip(legs, row)]) for row in inds]
    return TensorSpec(tuple(moduli), list(legs), inds, blocks, num_codomain)


def flip(leg: LegSpec) -> LegSpec:
    return LegSpec(leg.sectors, leg.mults, -leg.sign)


# --------------------------------------------------------------------------------------------- configs

def mps_pair(vleg: LegSpec, pleg: LegSpec, moduli, seed=DEFAULT_SEED):
    """Two MPS tensors A, B [vL(+), p(+), vR(-)] whose contraction over A.vR / B.vL is the
    two-site theta (cfg2 / cfg3 / the headline metric)."""
    rng = np.random.default_rng(seed)
    legs = [vleg, pleg, flip(vleg)]
    A = random_tensor(moduli, legs, rng, num_codomain=2)
    B = random_tensor(moduli, legs, rng, num_codomain=1)
    return A, B


def config_z2_chi64(seed=DEFAULT_SEED):
    """cfg1: Z2, 2-leg tensors over leg {0:32, 1:32}."""
    rng = np.random.default_rng(seed)
    leg = make_leg((2,), [[0], [1]], [32, 32], +1)
    A = random_tensor((2,), [leg, flip(leg)], rng, num_codomain=1)
    B = random_tensor((2,), [leg, flip(leg)], rng, num_codomain=1)
    return A, B


def config_u1_mps(chi=1024, seed=DEFAULT_SEED):
    """cfg2 (chi=1024) and the headline metric's workload (chi=4096): U(1) MPS two-site theta."""
    v = u1_leg(chi, 2.0)
    p = make_leg((0,), [[-1], [1]], [1, 1], +1)
    return mps_pair(v, p, (0,), seed)


def config_u1u1_mps(chi=4096, seed=DEFAULT_SEED):
    """cfg3: U(1) x U(1) (charge + 2Sz) theta."""
    v = u1u1_leg(chi, 2.0, 1.5)
    p = make_leg((0, 0), [[0, 0], [1, 1], [1, -1], [2, 0]], [1, 1, 1, 1], +1)
    return mps_pair(v, p, (0, 0), seed)


def config_su2_gemm_list(chi=512, seed=DEFAULT_SEED):
    """cfg4: the per-coupled-sector GEMM list of a FusionTreeBackend compose of two 3-leg SU(2)
    tensors (fusion_tree_backend.cpp:669-698: one matrix_dot per coupled spin, no accumulation).
    Multiplicities from the Gaussian recipe over 2j with sum m_j (2j+1) ~ chi.  Returns a list of
    (rows, K, cols) shapes and the numpy operands."""
    rng = np.random.default_rng(seed)
    twoj = np.arange(0, 13)
    w = np.exp(-(twoj / 2.0) ** 2 / (2 * 1.5 ** 2))
    m = np.floor(chi * w / np.sum(w * (twoj + 1))).astype(int)
    m = np.maximum(m, 0)
    mult = {int(t): int(x) for t, x in zip(twoj, m) if x > 0}
    # fuse virtual leg with a spin-1/2 physical leg: degeneracy of coupled spin J is m_{J-1} + m_{J+1}
    fused = {}
    for t, x in mult.items():
        for J in (t - 1, t + 1):
            if J >= 0:
                fused[J] = fused.get(J, 0) + x
    shapes, ops = [], []
    for J, rows in sorted(fused.items()):
        K = mult.get(J, 0)
        if K == 0:
            continue
        cols = rows
        shapes.append((rows, K, cols))
        ops.append((rng.standard_normal((rows, K)), rng.standard_normal((K, cols))))
    return shapes, ops


def config_ctmrg_blocks(chi=256, D=6, seed=DEFAULT_SEED, scale=1.0):
    """cfg5: per-sector blocks of the CTMRG enlarged corner ((chi D^2)^2 hermitian -> eigh) and of
    the (chi D^2) x chi projector matrix (-> QR), U(1) sectors from the Gaussian recipe.
    `scale` < 1 shrinks every multiplicity (test sizes)."""
    rng = np.random.default_rng(seed)
    big = u1_leg(int(chi * D * D * scale), 3.0)
    small = u1_leg(max(int(chi * scale), 1), 2.0)
    small_m = {int(q[0]): int(m) for q, m in zip(small.sectors, small.mults)}
    herm, tall = [], []
    for q, m in zip(big.sectors, big.mults):
        a = rng.standard_normal((int(m), int(m)))
        herm.append((a + a.T) / 2)
        c = small_m.get(int(q[0]), 0)
        if c:
            tall.append(rng.standard_normal((int(m), c)))
    return herm, tall


def theta_flops(A: TensorSpec, B: TensorSpec):
    """Algorithmic GEMM flops/bytes of contracting A's last leg with B's first (2MNK, 8(MK+KN+MN))."""
    # group B blocks by their first-leg sector
    flops = 0.0
    out_elems = {}
    bytes_ = 0.0
    b_by = {}
    for row, blk in zip(B.block_inds, B.blocks):
        b_by.setdefault(int(row[0]), []).append((row, blk))
    for row, blk in zip(A.block_inds, A.blocks):
        M = int(np.prod(blk.shape[:-1]))
        K = blk.shape[-1]
        for brow, bblk in b_by.get(int(row[-1]), []):
            N = int(np.prod(bblk.shape[1:]))
            flops += 2.0 * M * N * K
            bytes_ += 8.0 * (M * K + K * N)
            out_elems[(tuple(row[:-1]), tuple(brow[1:]))] = M * N
    bytes_ += 8.0 * sum(out_elems.values())
    return flops, bytes_, len(out_elems)


def svd_nominal_flops(shapes):
    """4 m n^2 + 8 n^3 with n = min, m = max per block (Golub-Van Loan R-SVD count)."""
    tot = 0.0
    for m, n in shapes:
        lo, hi = min(m, n), max(m, n)
        tot += 4.0 * hi * lo * lo + 8.0 * lo ** 3
    return tot


def config_heff(chi=256, D=5, seed=DEFAULT_SEED, charged_mpo=False, hermitian=True):
    """Tensors of a two-site effective Hamiltonian (leg orders of
    ``cyten_amd.krylov``): U(1) virtual legs from :func:`u1_leg`, spin-1/2-like physical legs, an MPO
    bond of dimension D.

    ``charged_mpo=False``: the MPO bond carries charge 0 only and every factor is symmetrised, so that
    H_eff is a sum of Kronecker products of symmetric matrices -- Hermitian by construction (what
    Lanczos needs).  ``charged_mpo=True``: bond sectors {-2, 0, 2} (hopping-like terms), random
    blocks, in general NOT Hermitian: for matvec parity and timing only.
    Returns dict(LP, W1, W2, RP, theta) of TensorSpec."""
    rng = np.random.default_rng(seed)
    mod = (0,)
    v = u1_leg(chi, 2.0)
    p = make_leg(mod, [[-1], [1]], [1, 1], +1)
    if charged_mpo:
        w = make_leg(mod, [[-2], [0], [2]], [1, max(D - 2, 1), 1], +1)
    else:
        w = make_leg(mod, [[0]], [D], +1)
    theta = random_tensor(mod, [v, p, p, flip(v)], rng, num_codomain=2)
    LP = random_tensor(mod, [v, w, flip(v)], rng, num_codomain=2)
    W1 = random_tensor(mod, [p, w, flip(p), flip(w)], rng, num_codomain=2)
    W2 = random_tensor(mod, [p, w, flip(p), flip(w)], rng, num_codomain=2)
    RP = random_tensor(mod, [flip(w), v, flip(v)], rng, num_codomain=2)
    if hermitian and not charged_mpo:
        # LP[a, l, a'] and RP[r, b', b] are block diagonal in the virtual charge: symmetrise every block
        for i, blk in enumerate(LP.blocks):
            LP.blocks[i] = 0.5 * (blk + blk.transpose(2, 1, 0))
        for i, blk in enumerate(RP.blocks):
            RP.blocks[i] = 0.5 * (blk + blk.transpose(0, 2, 1))
        # W[p', c, p, l]: p' = p (1 x 1 in the physical indices), nothing to symmetrise
    return dict(LP=LP, W1=W1, W2=W2, RP=RP, theta=theta)
