"""End-to-end parity on the BASELINE configs: block-sparse theta tdot + SVD through the HIP path
vs the oracle's one-block-at-a-time restatement of the reference, same seeded inputs."""
import numpy as np
import pytest

from cyten_amd import abelian as ab
from cyten_amd import workloads as wl
from helpers import check_svd_invariants, to_device_tensor
from oracle import abelian_ref as ref

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _theta_case(bb, A, B, chi_max):
    oracle = ref.theta_tdot_svd(A, B, chi_max=chi_max)
    a, b = to_device_tensor(bb, A), to_device_tensor(bb, B)
    theta = ab.compose(bb, a, b, 1)
    # tdot: element-wise (test_tdot of the reference compares to 12 decimals)
    np.testing.assert_array_equal(theta.block_inds, oracle['theta_block_inds'])
    scale = max(np.abs(x).max() for x in oracle['theta_blocks'])
    for got, want in zip(theta.blocks, oracle['theta_blocks']):
        assert np.abs(bb.to_numpy(got) - want).max() <= TOL * scale
    # dense check too (what test_tdot does)
    np.testing.assert_allclose(theta.to_dense(bb), ref.to_dense(oracle['theta']), rtol=0, atol=TOL * scale)
    # combine + SVD: invariants per sector + singular values vs LAPACK
    mv = ab.combine_legs_to_matrix(bb, theta, 2)
    assert [tuple(c) for c in mv.charges] == [tuple(c) for c in oracle['charges']]
    U, S, Vh = ab.svd(bb, mv)
    for got, want in zip(mv.blocks, oracle['matrices']):
        assert np.abs(bb.to_numpy(got) - want).max() <= TOL * scale
    for m, u, s, vh, (_, sref, _) in zip(oracle['matrices'], U, S, Vh, oracle['usv']):
        check_svd_invariants(m, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh), TOL, sref=sref)
    # truncation: same kept set size, same error / norm
    mv2, Ut, St, Vt, err, new_norm = ab.truncated_svd(bb, theta, 2, chi_max=chi_max)
    assert sum(s.size for s in St) == int(oracle['mask'].sum())
    tot = oracle['err'] + oracle['new_norm']
    assert abs(err - oracle['err']) <= TOL * tot and abs(new_norm - oracle['new_norm']) <= TOL * tot
    assert abs(ab.norm(bb, theta) ** 2 - tot) <= TOL * tot
    # truncated factors reproduce the best rank-chi approximation
    for m, u, s, vh in zip(oracle['matrices'], Ut, St, Vt):
        u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
        assert u.shape[1] == len(s) == vh.shape[0]
    resid2 = sum(np.linalg.norm(m - (bb.to_numpy(u) * bb.to_numpy(s)) @ bb.to_numpy(vh)) ** 2
                 for m, u, s, vh in zip(oracle['matrices'], Ut, St, Vt))
    assert abs(resid2 - oracle['err']) <= 1e-9 * tot
    # split_legs of the truncated factors: bit-exact sub-blocks (row slices are views, column slices one batched gather)
    for side, blocks, maps in (('rows', Ut, mv2.row_maps), ('cols', Vt, mv2.col_maps)):
        for sec, idx, blk in ab.split_matrix_legs(bb, mv2, blocks, side):
            off = [o for i, o, s_ in maps[sec] if i == idx][0]
            full, part = bb.to_numpy(blocks[sec]), bb.to_numpy(blk)
            if side == 'rows':
                rows = int(np.prod(part.shape[:-1]))
                np.testing.assert_array_equal(part.reshape(rows, -1), full[off:off + rows])
            else:
                cols = int(np.prod(part.shape[1:]))
                np.testing.assert_array_equal(part.reshape(part.shape[0], cols), full[:, off:off + cols])


def test_cfg1_z2_chi64(bb):
    A, B = wl.config_z2_chi64()
    oracle_blocks, bi, _ = ref.compose(A, B, 1)
    out = ab.compose(bb, to_device_tensor(bb, A), to_device_tensor(bb, B), 1)
    np.testing.assert_array_equal(out.block_inds, bi)
    for got, want in zip(out.blocks, oracle_blocks):
        assert np.abs(bb.to_numpy(got) - want).max() <= TOL * np.abs(want).max()
    mv = ab.combine_legs_to_matrix(bb, out, 1)
    U, S, Vh = ab.svd(bb, mv)
    for m, u, s, vh in zip(oracle_blocks, U, S, Vh):
        check_svd_invariants(m, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh), TOL)


def test_cfg2_u1_chi256(bb):
    _theta_case(bb, *wl.config_u1_mps(256), chi_max=256)


def test_cfg2_u1_chi1024(bb):
    _theta_case(bb, *wl.config_u1_mps(1024), chi_max=1024)


def test_cfg3_u1u1_downscaled(bb):
    """cfg3 at chi=512 (the full chi=4096 block list is covered by the bench + properties)."""
    _theta_case(bb, *wl.config_u1u1_mps(512), chi_max=300)


def test_cfg4_su2_gemm_list(bb):
    """FusionTreeBackend::compose = one matrix_dot per coupled sector (fusion_tree_backend.cpp:685)."""
    shapes, operands = wl.config_su2_gemm_list(512)
    outs = bb.matrix_dot_grouped([[(bb.as_block(a), bb.as_block(b))] for a, b in operands])
    for o, (a, b) in zip(outs, operands):
        assert np.abs(bb.to_numpy(o) - a @ b).max() <= TOL * np.abs(a @ b).max()


def test_cfg5_ctmrg_eigh_qr_downscaled(bb):
    herm, tall = wl.config_ctmrg_blocks(scale=0.05)
    for m, (W, V) in zip(herm, bb.eigh_batched([bb.as_block(h) for h in herm])):
        W, V = bb.to_numpy(W), bb.to_numpy(V)
        wref = np.linalg.eigvalsh(m)
        s = np.abs(wref).max()
        assert np.abs(W - wref).max() <= TOL * s and np.abs(m @ V - V * W).max() <= TOL * s
        assert np.abs(V.T @ V - np.eye(len(W))).max() <= TOL
    for m, (Q, R) in zip(tall, bb.matrix_qr_batched([bb.as_block(t) for t in tall])):
        Q, R = bb.to_numpy(Q), bb.to_numpy(R)
        assert np.abs(Q @ R - m).max() <= TOL * np.abs(m).max()
        assert np.abs(Q.T @ Q - np.eye(Q.shape[1])).max() <= TOL


def test_two_leg_contraction_and_inner(bb, rng):
    mod = (0,)
    v = wl.u1_leg(30, 1.5)
    p = wl.make_leg(mod, [[-1], [1]], [2, 3], +1)
    A = wl.random_tensor(mod, [v, wl.flip(p), wl.flip(v)], rng)
    B = wl.random_tensor(mod, [v, p, wl.flip(v)], rng)
    a, b = to_device_tensor(bb, A), to_device_tensor(bb, B)
    out = ab.compose(bb, a, b, 2)
    blocks, bi, _ = ref.compose(A, B, 2)
    np.testing.assert_array_equal(out.block_inds, bi)
    for got, want in zip(out.blocks, blocks):
        assert np.abs(bb.to_numpy(got) - want).max() <= TOL * max(1.0, np.abs(want).max())
    dense = ref.to_dense(B)
    assert abs(ab.inner(bb, b, b) - np.sum(dense * dense)) <= 1e-10 * np.sum(dense * dense)


def test_tdot_arbitrary_legs(bb):
    """cyten.tdot(a, b, legs_a, legs_b) = two leg permutations (views) + one grouped-GEMM compose."""
    A, B = wl.config_u1_mps(96)
    a, b = to_device_tensor(bb, A), to_device_tensor(bb, B)
    da, db = a.to_dense(bb), b.to_dense(bb)
    t = ab.tdot(bb, a, b, [2], [0])
    ref_ = np.tensordot(da, db, axes=([2], [0]))
    assert np.abs(t.to_dense(bb) - ref_).max() <= TOL * np.abs(ref_).max()
    Bf = wl.random_tensor(B.moduli, [B.legs[0], wl.flip(B.legs[1]), B.legs[2]], np.random.default_rng(5), num_codomain=1)
    bf = to_device_tensor(bb, Bf)
    t2 = ab.tdot(bb, a, bf, [2, 1], [0, 1])
    ref2 = np.tensordot(da, bf.to_dense(bb), axes=([2, 1], [0, 1]))
    assert np.abs(t2.to_dense(bb) - ref2).max() <= TOL * np.abs(ref2).max()


def test_device_truncation_matches_host_selection(bb, rng):
    """cyb_truncate_select_f64 against the host restatement of _truncate_singular_values_selection
    (tensor_backend.cpp:139-242): identical masks (ties included: both order equal values by position), err and
    new_norm to rounding, for every combination of the options, degenerate multiplets and constraint sets that
    contradict each other (the contradicting constraint is ignored, as combine_constraints does)."""
    for trial in range(60):
        n_sec = int(rng.integers(1, 9))
        sizes = [int(rng.integers(1, 70)) for _ in range(n_sec)]
        S = [np.sort(np.abs(rng.standard_normal(k)) * 10.0 ** rng.integers(-8, 1))[::-1].copy() for k in sizes]
        if trial % 3 == 0:   # exact multiplets across and inside sectors
            for s in S:
                s[:] = np.round(s / s.max(), 1) * s.max()
            S[-1][:] = S[0][0]
        if trial % 7 == 0:
            S[0][-1] = 0.0
        n = sum(sizes)
        opts = dict(chi_max=rng.choice([None, 1, max(1, n // 3), n, n + 5]), chi_min=int(rng.choice([1, 2, 5, n + 3])),
                    degeneracy_tol=float(rng.choice([0.0, 1e-8, 0.3])), trunc_cut=float(rng.choice([0.0, 1e-9, 1e-3, 1e3])),
                    svd_min=rng.choice([None, 1e-10, 0.5]), minimize_error=bool(rng.integers(0, 2)))
        opts['chi_max'] = None if opts['chi_max'] is None else int(opts['chi_max'])
        opts['svd_min'] = None if opts['svd_min'] is None else float(opts['svd_min'])
        want_mask, want_err, want_norm = ab.truncation_selection(np.concatenate(S), **opts)
        tables, mask, err, new_norm = bb.truncate_select([bb.as_block(s) for s in S], **opts)
        np.testing.assert_array_equal(bb.to_numpy(mask), want_mask, err_msg=f'trial {trial}: {opts}')
        tot = want_err + want_norm
        assert abs(err - want_err) <= 1e-13 * tot and abs(new_norm - want_norm) <= 1e-13 * tot
        offs = np.concatenate([[0], np.cumsum(sizes)])
        for s, t in enumerate(tables):
            assert t.n == int(want_mask[offs[s]:offs[s + 1]].sum())
        # the device tables drive the gather directly
        got = bb.mask_gather_many([(bb.as_block(s), t, 0) for s, t in zip(S, tables)])
        for s, g, k in zip(S, got, range(n_sec)):
            np.testing.assert_array_equal(bb.to_numpy(g), s[want_mask[offs[k]:offs[k + 1]]])
    with pytest.raises(NotImplementedError):
        bb.truncate_select([bb.as_block(np.ones(70000))])
    with pytest.raises(ValueError):
        bb.truncate_select([])


@pytest.mark.parametrize('n_total', [8193, 12000, 16384, 30000, 65536])
def test_device_truncation_beyond_one_lds_sort(bb, rng, n_total):
    """More than 8192 values (the U(1)xU(1) chi=4096 theta has ~16k): the sort runs in 8192-value chunks through LDS with
    the long-distance steps in device memory -- same masks, bit for bit, as the host selection, incl. exact multiplets."""
    cuts = np.sort(rng.choice(np.arange(1, n_total), size=36, replace=False))
    sizes = np.diff(np.concatenate([[0], cuts, [n_total]])).astype(int)
    S = [np.sort(np.abs(rng.standard_normal(k)) * 10.0 ** rng.integers(-6, 1))[::-1].copy() for k in sizes]
    S[3][:] = np.round(S[3] / S[3].max(), 2) * S[3].max()      # multiplets
    S[5][:min(len(S[5]), len(S[3]))] = S[3][:min(len(S[5]), len(S[3]))]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    for opts in (dict(chi_max=n_total // 2), dict(chi_max=4096, svd_min=1e-7, degeneracy_tol=1e-6),
                 dict(chi_max=None, trunc_cut=1e-2), dict(chi_max=n_total - 1, chi_min=n_total // 3, minimize_error=False)):
        want_mask, want_err, want_norm = ab.truncation_selection(np.concatenate(S), **opts)
        tables, mask, err, new_norm = bb.truncate_select([bb.as_block(s) for s in S], **opts)
        np.testing.assert_array_equal(bb.to_numpy(mask), want_mask, err_msg=str(opts))
        tot = want_err + want_norm
        assert abs(err - want_err) <= 1e-12 * tot and abs(new_norm - want_norm) <= 1e-12 * tot
        for s, t in enumerate(tables):
            assert t.n == int(want_mask[offs[s]:offs[s + 1]].sum())
    got = bb.mask_gather_many([(bb.as_block(s), t, 0) for s, t in zip(S, tables)])
    for k, (s, g) in enumerate(zip(S, got)):
        np.testing.assert_array_equal(bb.to_numpy(g), s[want_mask[offs[k]:offs[k + 1]]])


@pytest.mark.parametrize('chi_max', [100, 4000])
def test_truncated_svd_with_lazy_null_vectors_equals_the_eager_path(bb, chi_max):
    """``truncated_svd(lazy_null=True)``: the completion of the null vectors of rank-deficient sectors is skipped unless
    the truncation keeps them (chi_max = 4000 keeps every value, the zero ones included: those sectors are redone).
    Same S, same error / norm, same truncated product, orthonormal kept factors."""
    A, B = wl.config_u1_mps(192)
    a, b = ab.AbelianTensor.from_spec(bb, A), ab.AbelianTensor.from_spec(bb, B)
    theta = ab.compose(bb, a, b, 1)
    mv, U0, S0, V0, err0, nn0 = ab.truncated_svd(bb, theta, 2, chi_max=chi_max)
    mv1, U1, S1, V1, err1, nn1 = ab.truncated_svd(bb, theta, 2, lazy_null=True, chi_max=chi_max)
    assert abs(err0 - err1) <= 1e-10 * (err0 + nn0) and abs(nn0 - nn1) <= 1e-10 * nn0
    for m, u0, s0, v0, u1, s1, v1 in zip(mv.blocks, U0, S0, V0, U1, S1, V1):
        s0n, s1n = bb.to_numpy(s0), bb.to_numpy(s1)
        assert s0n.shape == s1n.shape
        nrm = max(np.linalg.norm(bb.to_numpy(m)), 1e-300)
        assert np.abs(s0n - s1n).max(initial=0.0) <= TOL * nrm
        u1n, v1n = bb.to_numpy(u1), bb.to_numpy(v1)
        p0 = (bb.to_numpy(u0) * s0n) @ bb.to_numpy(v0)
        assert np.abs((u1n * s1n) @ v1n - p0).max(initial=0.0) <= TOL * nrm
        k = len(s1n)
        assert np.abs(u1n.T @ u1n - np.eye(k)).max(initial=0.0) <= TOL and np.abs(v1n @ v1n.T - np.eye(k)).max(initial=0.0) <= TOL
    # the C-ABI form: ranks of the theta-like blocks are about half their size, unknown flag bits are refused
    res, ranks = bb.matrix_svd_batched(mv.blocks, null_vectors=False, return_rank=True)
    big = [(blk.shape, r) for blk, r in zip(mv.blocks, ranks) if min(blk.shape) >= 48]
    assert big and all(r < min(shp) for shp, r in big)
