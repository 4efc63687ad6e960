"""Pin the CPU oracle with the reference's own known-answer checks for this path.

The reference holds no golden vectors for tdot/SVD/QR/eigh; its tests pin results through
properties (tests/python_tests/test_tensors.py: test_tdot :3616-3765 -- dense np.tensordot to 12
decimals; test_svd :3393-3500; test_qr_lq :3166; test_eigh :2057).  The same checks are applied
to ``oracle/`` here, on the BASELINE config generators at CPU-sized chi.
"""
import numpy as np
import pytest

from cyten_amd import workloads as wl
from oracle import abelian_ref as ref
from oracle import block_ops as ops


def _dense_theta(A, B, num_contr=1):
    da, db = ref.to_dense(A), ref.to_dense(B)
    axa = list(range(da.ndim - 1, da.ndim - 1 - num_contr, -1))
    axb = list(range(num_contr))
    return np.tensordot(da, db, (axa, axb))


@pytest.mark.parametrize('maker', [
    lambda: wl.config_z2_chi64(),
    lambda: wl.config_u1_mps(64),
    lambda: wl.config_u1_mps(256, seed=7),
    lambda: wl.config_u1u1_mps(200),
])
def test_compose_matches_dense_tensordot(maker):
    """test_tdot :3757-3765: block-sparse result == np.tensordot of the dense arrays, 12 decimals."""
    A, B = maker()
    blocks, bi, n_dot = ref.compose(A, B, 1)

    class T:
        moduli, legs, block_inds = A.moduli, list(A.legs[:-1]) + list(B.legs[1:]), bi
    T.blocks = blocks
    np.testing.assert_almost_equal(ref.to_dense(T), _dense_theta(A, B), decimal=12)
    assert n_dot >= len(blocks)


def test_compose_two_contracted_legs(rng):
    """Contracting two legs exercises the reversed-order key packing (abelian.cpp:1265-1283)."""
    mod = (0,)
    v = wl.u1_leg(40, 1.5)
    p = wl.make_leg(mod, [[-1], [1]], [2, 3], +1)
    A = wl.random_tensor(mod, [v, wl.flip(p), wl.flip(v)], rng)        # contract legs (p-, v-)
    B = wl.random_tensor(mod, [v, p, wl.flip(v)], rng)                 # with B's (v+, p+)
    blocks, bi, _ = ref.compose(A, B, 2)

    class T:
        moduli, legs, block_inds = mod, [A.legs[0], B.legs[2]], bi
    T.blocks = blocks
    np.testing.assert_almost_equal(ref.to_dense(T), _dense_theta(A, B, 2), decimal=12)


def test_compose_sparse_and_empty(rng):
    mod = (3,)
    leg = wl.make_leg(mod, [[0], [1], [2]], [3, 1, 4], +1)
    A = wl.random_tensor(mod, [leg, wl.flip(leg)], rng, fill=0.6)
    B = wl.random_tensor(mod, [leg, wl.flip(leg)], rng, fill=0.6)
    blocks, bi, _ = ref.compose(A, B, 1)

    class T:
        moduli, legs, block_inds = mod, [A.legs[0], B.legs[1]], bi
    T.blocks = blocks
    np.testing.assert_almost_equal(ref.to_dense(T), _dense_theta(A, B), decimal=12)
    E = wl.TensorSpec(mod, [leg, wl.flip(leg)], np.zeros((0, 2), np.int64), [])
    assert ref.compose(A, E, 1)[0] == [] and ref.compose(E, B, 1)[0] == []


def test_theta_svd_properties():
    """test_svd :3405-3500 on the combined theta: per-sector U S Vh = block, isometries, |S| = |T|,
    and truncated_svd's err == |T - T_approx|^2 contribution."""
    A, B = wl.config_u1_mps(128)
    res = ref.theta_tdot_svd(A, B, chi_max=40)
    dense = _dense_theta(A, B)
    assert abs(np.linalg.norm(res['S_all']) - np.linalg.norm(dense)) < 1e-10 * np.linalg.norm(dense)
    for m, (U, S, Vh) in zip(res['matrices'], res['usv']):
        assert np.all(S >= 0) and np.all(np.diff(S) <= 1e-12)
        np.testing.assert_allclose((U * S) @ Vh, m, atol=1e-10)
        np.testing.assert_allclose(U.T @ U, np.eye(len(S)), atol=1e-10)
        np.testing.assert_allclose(Vh @ Vh.T, np.eye(len(S)), atol=1e-10)
    mask, err, new_norm = res['mask'], res['err'], res['new_norm']
    assert mask.sum() == 40
    # err is the discarded weight: |T - T_approx|^2 = sum of discarded S^2
    offs = np.cumsum([0] + [len(s) for _, s, _ in res['usv']])
    resid2 = 0.0
    for i, (m, (U, S, Vh)) in enumerate(zip(res['matrices'], res['usv'])):
        keep = mask[offs[i]:offs[i + 1]]
        resid2 += np.linalg.norm(m - (U[:, keep] * S[keep]) @ Vh[keep]) ** 2
    assert abs(resid2 - err) < 1e-10 * (err + new_norm)
    assert abs(err + new_norm - np.linalg.norm(dense) ** 2) < 1e-9 * np.linalg.norm(dense) ** 2


def test_truncation_selection_options():
    S = np.array([0.9, 0.3, 0.3 * (1 + 1e-12), 0.1, 1e-9, 0.0])
    mask, err, nn = ref.truncation_selection(S, chi_max=3, degeneracy_tol=1e-6)
    # cannot cut between the two (numerically) degenerate values -> keeps only 0.9 ... or all of them
    assert mask.sum() in (1, 3) and mask[0]
    mask, err, nn = ref.truncation_selection(S, svd_min=1e-6)
    assert list(mask) == [True, True, True, True, False, False]
    assert abs(err - (1e-18)) < 1e-24
    mask, _, _ = ref.truncation_selection(S, chi_max=2, chi_min=2)
    assert mask.sum() == 2
    mask, err, nn = ref.truncation_selection(np.array([3.0, 4.0]), trunc_cut=3.5)
    assert list(mask) == [False, True] and err == 9.0 and nn == 16.0


def test_qr_lq_eigh_properties(rng):
    """test_qr_lq :3166 and test_eigh :2057 property checks on the per-block oracle ops."""
    for m, n in [(7, 4), (4, 7), (12, 12)]:
        a = rng.standard_normal((m, n))
        for full in (False, True):
            q, r = ops.matrix_qr(a, full)
            np.testing.assert_allclose(q @ r, a, atol=1e-12)
            np.testing.assert_allclose(q.T @ q, np.eye(q.shape[1]), atol=1e-12)
            assert np.abs(np.tril(r, -1)).max() < 1e-14
            l, q2 = ops.matrix_lq(a, full)
            np.testing.assert_allclose(l @ q2, a, atol=1e-12)
            np.testing.assert_allclose(q2 @ q2.T, np.eye(q2.shape[0]), atol=1e-12)
    h = rng.standard_normal((9, 9))
    h = h + h.T
    for sort in (None, 'm>', '<', 'LM'):
        w, v = ops.eigh(h, sort)
        np.testing.assert_allclose(h @ v, v * w, atol=1e-11)
        np.testing.assert_allclose(v.T @ v, np.eye(9), atol=1e-12)
    w, _ = ops.eigh(h, 'm>')
    assert np.all(np.diff(np.abs(w)) <= 1e-14)


def test_block_ops_semantics(rng):
    a = rng.standard_normal((3, 4, 5))
    f = rng.standard_normal(4)
    np.testing.assert_allclose(ops.scale_axis(a, f, 1), a * f[None, :, None])
    mask = np.array([True, False, True, True])
    np.testing.assert_array_equal(ops.apply_mask(a, mask, 1), a[:, mask])
    np.testing.assert_array_equal(ops.apply_mask(ops.enlarge_leg(a[:, mask], mask, 1), mask, 1), a[:, mask])
    np.testing.assert_allclose(ops.combine_legs(a, [[0, 1]]), a.reshape(12, 5))
    np.testing.assert_allclose(ops.combine_legs(a, [[1, 2]], cstyles=False), a.transpose(0, 2, 1).reshape(3, 20))
    with pytest.raises(ValueError):
        ops.matrix_svd(a[0], 'nonsense')
