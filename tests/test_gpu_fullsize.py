"""BASELINE-size checks of the decomposition path: size-independent properties evaluated ON the device AND the values of
the oracle's LAPACK routines (`oracle.block_ops.matrix_svd` / `eigh` = the scipy / numpy calls of numpy.cpp:1247-1297,
:658-698) on the same blocks -- all 15 coupled-charge blocks of the chi = 4096 U(1) theta (up to 1442 x 1442), the largest
blocks of the U(1)xU(1) list and the cfg5 eigenvalues; fp64 tolerance 1e-10 relative to the block norm."""
import numpy as np
import pytest

from oracle import block_ops as ops
from cyten_amd import abelian as ab
from cyten_amd import workloads as wl
from helpers import to_device_tensor

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_chi4096_theta_svd_properties(bb):
    A, B = wl.config_u1_mps(4096)
    a, b = to_device_tensor(bb, A), to_device_tensor(bb, B)
    theta = ab.compose(bb, a, b, 1)
    mv = ab.combine_legs_to_matrix(bb, theta, 2)
    shapes = [tuple(m.shape) for m in mv.blocks]
    assert len(shapes) == 15 and max(max(s) for s in shapes) == 1442
    U, S, Vh = ab.svd(bb, mv)
    # (i) reconstruction  U diag(S) Vh == M, one grouped GEMM for all sectors
    US = bb.scale_axis_many([(u, s, 1) for u, s in zip(U, S)])
    rec = bb.matrix_dot_grouped([[(us, vh)] for us, vh in zip(US, Vh)])
    diff = bb.linear_combination_many(1.0, rec, -1.0, mv.blocks)
    total2 = 0.0
    for d, m, s in zip(diff, mv.blocks, S):
        nrm = bb.norm(m)
        assert bb.max_abs(d) <= TOL * nrm
        # (ii) sum of squared singular values == squared Frobenius norm
        s_np = bb.to_numpy(s)
        assert abs(np.sum(s_np ** 2) - nrm ** 2) <= TOL * nrm ** 2
        # (iii) descending, non-negative
        assert np.all(s_np >= 0) and np.all(np.diff(s_np) <= 1e-12 * s_np[0])
        total2 += nrm ** 2
    # (iv) isometries: U^T U = I, Vh Vh^T = I (grouped GEMMs on transposed views)
    gram_u = bb.matrix_dot_grouped([[(bb.permute_axes(u, [1, 0]), u)] for u in U])
    gram_v = bb.matrix_dot_grouped([[(vh, bb.permute_axes(vh, [1, 0]))] for vh in Vh])
    for g in gram_u + gram_v:
        eye = bb.eye_matrix(g.shape[0])
        assert bb.max_abs(bb.linear_combination(1.0, g, -1.0, eye)) <= TOL
    # (v) every block of theta = A.B factors through ONE sector of the shared bond: its numerical rank is
    #     that sector's multiplicity (random blocks: exactly), about half the block size
    bond_mults = set(int(m) for m in wl.u1_leg(4096, 2.0).mults)
    for s, shp in zip(S, shapes):
        s_np = bb.to_numpy(s)
        rank = int(np.sum(s_np > 1e-9 * s_np[0]))
        assert rank in bond_mults and rank <= min(shp)
        assert np.all(s_np[rank:] <= 1e-10 * s_np[0])
    # (vi) the singular values of LAPACK (the oracle's per-block loop, = the reference's call) on the same blocks, all 15
    for m, s in zip(mv.blocks, S):
        m_np = bb.to_numpy(m)
        _, s_ref, _ = ops.matrix_svd(m_np)
        assert np.abs(bb.to_numpy(s) - s_ref).max() <= TOL * np.linalg.norm(m_np)
    # (vii) norm bookkeeping of the truncation at chi_max = 4096
    _, Ut, St, Vt, err, new_norm = ab.truncated_svd(bb, theta, 2, chi_max=4096)
    assert sum(s.size for s in St) == 4096
    assert abs(err + new_norm - total2) <= TOL * total2


def test_u1u1_chi4096_theta_svd_properties(bb):
    """cfg3 (the north-star list): U(1)xU(1) chi = 4096 theta -- 728 GEMMs with K-split accumulation, 59 coupled-charge
    blocks up to 1351 x 1351, more than 8192 singular values in the device truncation -- through the same
    size-independent properties, evaluated on the device."""
    A, B = wl.config_u1u1_mps(4096)
    a, b = to_device_tensor(bb, A), to_device_tensor(bb, B)
    theta = ab.compose(bb, a, b, 1)
    assert len(theta.blocks) > 500
    # linearity of the contraction in its first operand (one more compose): (2a) . b == 2 (a . b)
    theta2 = ab.compose(bb, ab.scale(bb, 2.0, a), b, 1)
    diff = bb.linear_combination_many(2.0, theta.blocks, -1.0, theta2.blocks)
    assert bb.max_abs_many(diff) <= 1e-12 * bb.max_abs_many(theta.blocks)
    mv = ab.combine_legs_to_matrix(bb, theta, 2)
    assert abs(bb.norm_many(mv.blocks) - ab.norm(bb, theta)) <= 1e-12 * ab.norm(bb, theta)   # combine is a permutation
    U, S, Vh = ab.svd(bb, mv)
    US = bb.scale_axis_many([(u, s, 1) for u, s in zip(U, S)])
    rec = bb.matrix_dot_grouped([[(us, vh)] for us, vh in zip(US, Vh)])
    diff = bb.linear_combination_many(1.0, rec, -1.0, mv.blocks)
    total2 = 0.0
    for d, m, s in zip(diff, mv.blocks, S):
        nrm = bb.norm(m)
        assert bb.max_abs(d) <= TOL * max(nrm, 1e-300)
        s_np = bb.to_numpy(s)
        assert abs(np.sum(s_np ** 2) - nrm ** 2) <= TOL * max(nrm ** 2, 1e-300)
        assert np.all(s_np >= 0) and np.all(np.diff(s_np) <= 1e-12 * max(s_np[0], 1e-300))
        total2 += nrm ** 2
    gram_u = bb.matrix_dot_grouped([[(bb.permute_axes(u, [1, 0]), u)] for u in U])
    gram_v = bb.matrix_dot_grouped([[(vh, bb.permute_axes(vh, [1, 0]))] for vh in Vh])
    for g in gram_u + gram_v:
        assert bb.max_abs(bb.linear_combination(1.0, g, -1.0, bb.eye_matrix(g.shape[0]))) <= TOL
    # LAPACK's singular values on the ten largest blocks of the list
    for i in sorted(range(len(S)), key=lambda i: -mv.blocks[i].shape[0] * mv.blocks[i].shape[1])[:10]:
        m_np = bb.to_numpy(mv.blocks[i])
        _, s_ref, _ = ops.matrix_svd(m_np)
        assert np.abs(bb.to_numpy(S[i]) - s_ref).max() <= TOL * np.linalg.norm(m_np)
    n_values = sum(s.size for s in S)
    assert 8192 < n_values <= bb.TRUNCATE_MAX                                # the chunked device selection is what runs
    _, Ut, St, Vt, err, new_norm = ab.truncated_svd(bb, theta, 2, chi_max=4096)
    assert sum(s.size for s in St) == 4096
    assert abs(err + new_norm - total2) <= TOL * total2
    # the kept values are the 4096 largest of the whole list
    all_s = np.sort(np.concatenate([bb.to_numpy(s) for s in S]))[::-1]
    kept = np.sort(np.concatenate([bb.to_numpy(s) for s in St]))[::-1]
    np.testing.assert_array_equal(kept, all_s[:4096])


def test_cfg5_ctmrg_eigh_qr_fullsize_properties(bb):
    """cfg5 at its full size (D = 6, chi = 256: hermitian sector blocks up to 1238 x 1238, tall blocks up to 1238 x 58):
    eigh and QR through properties evaluated on the device -- A V = V diag(w), V^T V = 1, trace(A) = sum w, ascending w;
    A = Q R, Q^T Q = 1, R upper triangular with |R_ii| = the column norms of the Gram-Schmidt residuals (checked via
    ||R||_F = ||A||_F)."""
    herm, tall = wl.config_ctmrg_blocks()
    assert max(h.shape[0] for h in herm) > 1000
    H = [bb.as_block(h) for h in herm]
    res = bb.eigh_batched(H)
    AV = bb.matrix_dot_grouped([[(h, v)] for h, (w, v) in zip(H, res)])
    VW = bb.scale_axis_many([(v, w, 1) for w, v in res])
    diff = bb.linear_combination_many(1.0, AV, -1.0, VW)
    gram = bb.matrix_dot_grouped([[(bb.permute_axes(v, [1, 0]), v)] for _, v in res])
    for h, hd, d, g, (w, v) in zip(herm, H, diff, gram, res):
        nrm = bb.norm(hd)
        assert bb.max_abs(d) <= TOL * nrm
        assert bb.max_abs(bb.linear_combination(1.0, g, -1.0, bb.eye_matrix(g.shape[0]))) <= TOL
        w_np = bb.to_numpy(w)
        assert np.all(np.diff(w_np) >= -1e-12 * nrm)
        assert abs(np.sum(w_np) - np.trace(h)) <= TOL * nrm * np.sqrt(len(w_np))
        assert abs(np.sum(w_np ** 2) - nrm ** 2) <= TOL * nrm ** 2          # sum of squared eigenvalues = ||A||_F^2
        assert np.abs(w_np - ops.eigvalsh(h)).max() <= TOL * nrm             # LAPACK's eigenvalues (numpy.cpp:682-698)
    T = [bb.as_block(t) for t in tall]
    qr = bb.matrix_qr_batched(T)
    QR = bb.matrix_dot_grouped([[(q, r)] for q, r in qr])
    diff = bb.linear_combination_many(1.0, QR, -1.0, T)
    gram = bb.matrix_dot_grouped([[(bb.permute_axes(q, [1, 0]), q)] for q, _ in qr])
    for t, d, g, (q, r) in zip(T, diff, gram, qr):
        nrm = bb.norm(t)
        assert bb.max_abs(d) <= TOL * nrm
        assert bb.max_abs(bb.linear_combination(1.0, g, -1.0, bb.eye_matrix(g.shape[0]))) <= TOL
        r_np = bb.to_numpy(r)
        assert np.abs(np.tril(r_np, -1)).max() == 0.0
        assert abs(np.linalg.norm(r_np) - nrm) <= TOL * nrm
        _, r_ref = ops.matrix_qr(bb.to_numpy(t), False)                      # scipy.linalg.qr: R entry-wise (same sign convention)
        assert np.abs(r_np - r_ref).max() <= TOL * nrm
