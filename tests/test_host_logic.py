"""CPU tests of the product's HOST logic (no GPU): sector matching, combine/split maps,
truncation selection, view algebra, LPT sharding -- checked against the oracle."""
import numpy as np
import pytest

from cyten_amd import abelian as ab
from cyten_amd import sharding
from cyten_amd import workloads as wl
from cyten_amd.block_backend import _c_strides, _nocopy_reshape_strides
from oracle import abelian_ref as ref

from helpers import to_device_tensor
from numpy_backend import NumpyGroupedBackend


@pytest.fixture
def nb():
    return NumpyGroupedBackend()


CONFIGS = [lambda: wl.config_z2_chi64(), lambda: wl.config_u1_mps(96), lambda: wl.config_u1u1_mps(150)]


@pytest.mark.parametrize('maker', CONFIGS)
def test_compose_plan_matches_oracle(nb, maker):
    """Same result blocks (indices, shapes, values) and the same number of matched pairs as the
    one-block-at-a-time restatement of abelian_compose_worker."""
    A, B = maker()
    a, b = to_device_tensor(nb, A), to_device_tensor(nb, B)
    plan = ab.compose_plan(a, b, 1)
    blocks, bi, n_dot = ref.compose(A, B, 1)
    np.testing.assert_array_equal(plan.res_block_inds, bi)
    assert [tuple(x.shape) for x in blocks] == [tuple(s) for s in plan.res_shapes]
    assert sum(len(g) for g in plan.pairs) == n_dot
    flops, _, nblk = wl.theta_flops(A, B)
    assert plan.flops == flops and nblk == len(blocks)
    out = ab.compose(nb, a, b, 1)
    for x, y in zip(out.blocks, blocks):
        np.testing.assert_allclose(x, y, rtol=0, atol=1e-12)


def test_compose_plan_two_legs(nb, rng):
    mod = (0,)
    v = wl.u1_leg(30, 1.5)
    p = wl.make_leg(mod, [[-1], [1]], [2, 3], +1)
    A = wl.random_tensor(mod, [v, wl.flip(p), wl.flip(v)], rng)
    B = wl.random_tensor(mod, [v, p, wl.flip(v)], rng)
    out = ab.compose(nb, to_device_tensor(nb, A), to_device_tensor(nb, B), 2)
    blocks, bi, _ = ref.compose(A, B, 2)
    np.testing.assert_array_equal(out.block_inds, bi)
    for x, y in zip(out.blocks, blocks):
        np.testing.assert_allclose(x, y, atol=1e-12)


def test_compose_rejects_mismatched_legs(nb, rng):
    mod = (0,)
    v, w = wl.u1_leg(20, 1.0), wl.u1_leg(24, 1.0)
    A = wl.random_tensor(mod, [v, wl.flip(v)], rng)
    B = wl.random_tensor(mod, [w, wl.flip(w)], rng)
    with pytest.raises(ValueError):
        ab.compose_plan(to_device_tensor(nb, A), to_device_tensor(nb, B), 1)
    B2 = wl.random_tensor(mod, [wl.flip(v), v], rng)  # same orientation on both contracted legs
    with pytest.raises(ValueError):
        ab.compose_plan(to_device_tensor(nb, A), to_device_tensor(nb, B2), 1)


@pytest.mark.parametrize('maker', CONFIGS[1:])
def test_combine_svd_truncate_matches_oracle(nb, maker):
    A, B = maker()
    theta = ab.compose(nb, to_device_tensor(nb, A), to_device_tensor(nb, B), 1)
    oracle = ref.theta_tdot_svd(A, B, chi_max=50)
    mv = ab.combine_legs_to_matrix(nb, theta, 2)
    assert [tuple(c) for c in mv.charges] == [tuple(c) for c in oracle['charges']]
    for x, y in zip(mv.blocks, oracle['matrices']):
        np.testing.assert_allclose(x, y, atol=1e-12)
    mv2, U, S, Vh, err, new_norm = ab.truncated_svd(nb, theta, 2, chi_max=50)
    assert abs(err - oracle['err']) <= 1e-12 * (1 + oracle['err'])
    assert abs(new_norm - oracle['new_norm']) <= 1e-12 * oracle['new_norm']
    assert sum(s.size for s in S) == 50
    # split views reproduce the sub-blocks of U
    parts = ab.split_matrix_legs(nb, mv2, U, 'rows')
    for sec, idx, blk in parts:
        off = [o for i, o, s in mv2.row_maps[sec] if i == idx][0]
        rows = int(np.prod(blk.shape[:-1]))
        np.testing.assert_array_equal(blk.reshape(rows, blk.shape[-1]), U[sec][off:off + rows])
    # ... and the column slices of Vh (one batched gather on the device backend)
    for sec, idx, blk in ab.split_matrix_legs(nb, mv2, Vh, 'cols'):
        off = [o for i, o, s in mv2.col_maps[sec] if i == idx][0]
        cols = int(np.prod(blk.shape[1:]))
        np.testing.assert_array_equal(blk.reshape(blk.shape[0], cols), Vh[sec][:, off:off + cols])
    with pytest.raises(ValueError):
        ab.split_matrix_legs(nb, mv2, U, 'row')


def test_truncation_selection_equals_oracle(rng):
    for _ in range(20):
        S = np.abs(rng.standard_normal(rng.integers(1, 40))) * 10.0 ** rng.integers(-12, 1)
        opts = dict(chi_max=int(rng.integers(1, 30)), degeneracy_tol=float(rng.choice([0.0, 1e-8])),
                    trunc_cut=float(rng.choice([0.0, 1e-6])), svd_min=rng.choice([None, 1e-10]))
        m1, e1, n1 = ab.truncation_selection(S, **opts)
        m2, e2, n2 = ref.truncation_selection(S, **opts)
        np.testing.assert_array_equal(m1, m2)
        assert e1 == e2 and n1 == n2


def test_nocopy_reshape_rule(rng):
    """View-or-copy decision must agree with numpy's."""
    base = rng.standard_normal((4, 6, 5, 2))
    cases = [((0, 1, 2, 3), (24, 10)), ((0, 1, 2, 3), (4, 60)), ((1, 0, 2, 3), (24, 10)), ((0, 2, 1, 3), (4, 30, 2)),
             ((3, 2, 1, 0), (2, 5, 24)), ((0, 1, 2, 3), (2, 2, 6, 10)), ((1, 0, 2, 3), (6, 4, 10))]
    for perm, new_shape in cases:
        v = base.transpose(perm)
        strides = tuple(s // 8 for s in v.strides)
        st = _nocopy_reshape_strides(v.shape, strides, list(new_shape))
        r = v.reshape(new_shape)
        is_view = np.shares_memory(r, base)
        assert (st is not None) == is_view, (perm, new_shape)
        if st is not None:
            assert tuple(s // 8 for s in r.strides) == st
    assert _c_strides((3, 4, 5)) == (20, 5, 1)


def test_lpt_and_pool_layout():
    costs = [36.0, 20, 20, 8, 8, 8, 1, 1, 1, 1]
    owner = sharding.lpt_assign(costs, 4)
    loads = [sum(c for c, o in zip(costs, owner) if o == r) for r in range(4)]
    assert max(loads) == 36.0 and sorted(owner.tolist()).count(owner[0]) == 1
    sizes = [100, 37, 64, 5, 5, 999, 1, 1, 1, 1]
    lay = sharding.make_layout(sizes, costs, 4)
    assert lay.total == 4 * lay.seg_len
    spans = sorted((int(o), int(o + s)) for o, s in zip(lay.offset, lay.sizes))
    assert all(a1 <= b0 for (_, a1), (b0, _) in zip(spans[:-1], spans[1:]))  # no overlap
    for u in range(len(sizes)):
        r = lay.owner[u]
        assert r * lay.seg_len <= lay.offset[u] and lay.offset[u] + sizes[u] <= (r + 1) * lay.seg_len
        assert lay.offset[u] % 32 == 0
    assert lay.imbalance(costs) >= 1.0
    one = sharding.make_layout(sizes, costs, 1)
    assert set(one.owner.tolist()) == {0}


def test_tdot_arbitrary_legs_matches_dense(nb, rng):
    """tdot over arbitrary leg pairs = permute + compose; checked against numpy.tensordot of the dense arrays."""
    A, B = wl.config_u1_mps(48)
    a, b = to_device_tensor(nb, A), to_device_tensor(nb, B)
    da, db = a.to_dense(nb), b.to_dense(nb)
    # A[vL, p, vR] . B[vL', p', vR']: contract A.vR with B.vL (the theta contraction) ...
    t = ab.tdot(nb, a, b, [2], [0])
    np.testing.assert_allclose(t.to_dense(nb), np.tensordot(da, db, axes=([2], [0])), atol=1e-12)
    # ... and a two-leg contraction over (vR, p) x (vL, p) after flipping B's physical leg to make it contractible
    Bf = wl.TensorSpec(B.moduli, [B.legs[0], wl.flip(B.legs[1]), B.legs[2]], B.block_inds, B.blocks, B.num_codomain)
    # charges must still be conserved with the flipped leg: rebuild the allowed blocks
    Bf = wl.random_tensor(B.moduli, Bf.legs, np.random.default_rng(5), num_codomain=1)
    bf = to_device_tensor(nb, Bf)
    t2 = ab.tdot(nb, a, bf, [2, 1], [0, 1])
    np.testing.assert_allclose(t2.to_dense(nb), np.tensordot(da, bf.to_dense(nb), axes=([2, 1], [0, 1])), atol=1e-11)
    with pytest.raises(ValueError):
        ab.tdot(nb, a, b, [2, 2], [0, 1])


def test_oracle_api_restatements_identities(rng):
    """The oracle's restatements of the small operator-API helpers against independent numpy formulations."""
    from oracle import block_ops as ops
    mask = rng.random(11) < 0.5
    mask[0] = True
    P = ops.block_from_mask(mask)
    x = rng.standard_normal(11)
    np.testing.assert_array_equal(P @ x, x[mask])               # the projector of apply_mask
    for big in range(11):
        col = P[:, big]
        for small in range(P.shape[0]):
            assert ops.get_block_mask_element(mask, big, small) == bool(col[small])
    a = rng.standard_normal((3, 4, 4, 3, 2))
    np.testing.assert_allclose(ops.trace_partial(a, [0, 1], [3, 2], [4]), np.einsum('abbac->c', a), atol=1e-13)
    m = rng.standard_normal((6, 20))
    np.testing.assert_array_equal(ops.permute_combined_idx(m, 0, [2, 3], [1, 0]),
                                  m.reshape(2, 3, 20).transpose(1, 0, 2).reshape(6, 20))
    np.testing.assert_array_equal(ops.permute_combined_matrix(m, [2, 3], [0, 1], [4, 5], [3, 2]),
                                  m.reshape(2, 3, 4, 5).transpose(0, 1, 3, 2).reshape(6, 20))
    u, v = rng.standard_normal((2, 3)), rng.standard_normal((4,))
    np.testing.assert_allclose(ops.tensor_outer(u, v, 1), np.einsum('ab,c->acb', u, v))
    np.testing.assert_array_equal(ops.cutoff_inverse(np.array([2.0, 1e-12, -4.0]), 1e-6), [0.5, 0.0, -0.25])
    np.testing.assert_array_equal(ops.stable_log(np.array([1.0, 1e-12, -4.0]), 1e-6), [0.0, 0.0, 0.0])


def test_oracle_transform_blocks_against_einsum(rng):
    """oracle.block_ops.transform_blocks (the call-by-call restatement of TreePairMapping::transform_tensor's block
    arithmetic) against an independent einsum formulation of one tree pair."""
    from oracle import block_ops as ops
    old = [rng.standard_normal((6, 20)), rng.standard_normal((12, 20))]
    dims1, dims2, idcs1, idcs2 = [2, 3], [4, 5], [3, 0], [1, 2]
    terms = [(0.7, 0, (0, 6), (0, 20)), (-1.3, 1, (6, 12), (0, 20))]
    new = ops.transform_blocks(old, [(11, 13)], [(0, (1, 11), (0, 12), dims1, idcs1, dims2, idcs2, terms)])[0]
    t = 0.7 * old[0] - 1.3 * old[1][6:12]
    want = np.einsum('abcd->dabc', t.reshape(2, 3, 4, 5)).reshape(10, 12)
    np.testing.assert_allclose(new[1:11, 0:12], want, atol=1e-15)
    assert not new[0].any() and not new[:, 12].any()


def test_native_compose_planner_equals_numpy_specification(rng):
    """cyb_compose_plan_create (csrc/abelian_plan.hip, host-only C++) against compose_plan_py: identical result rows,
    shapes, pair lists (order included) and flops on the BASELINE configs, on tensors with missing blocks, several
    contracted legs, Z_N x U(1) charges and fully contracted legs; same error for legs that do not match."""
    if not ab._native_planner():
        pytest.skip('libcyten_amd.so is not built')
    nbk = NumpyGroupedBackend()

    def same(a, b, k):
        p1, p2 = ab.compose_plan(a, b, k), ab.compose_plan_py(a, b, k)
        np.testing.assert_array_equal(p1.res_block_inds, p2.res_block_inds)
        assert p1.res_shapes == [tuple(int(x) for x in s) for s in p2.res_shapes]
        assert p1.pairs == p2.pairs
        assert abs(p1.flops - p2.flops) <= 1e-9 * max(1.0, p2.flops)

    for maker in CONFIGS:
        A, B = maker()
        same(to_device_tensor(nbk, A), to_device_tensor(nbk, B), 1)
    for trial in range(30):
        mod = [(0,), (3,), (0, 2), (4, 0)][trial % 4]
        nsym = len(mod)

        def leg(nsec, sign):
            secs = rng.integers(-3, 4, size=(nsec, nsym))
            secs = np.unique(ab.Symmetry(mod).reduce(secs), axis=0)
            return wl.make_leg(mod, secs.tolist(), rng.integers(1, 5, size=len(secs)).tolist(), sign)
        k = int(rng.integers(0, 3))
        contr = [leg(int(rng.integers(1, 5)), +1) for _ in range(k)]
        a_legs = [leg(int(rng.integers(1, 5)), int(rng.choice([-1, 1]))) for _ in range(int(rng.integers(0, 3)))] + contr[::-1]
        b_legs = [wl.flip(c) for c in contr] + [leg(int(rng.integers(1, 5)), int(rng.choice([-1, 1]))) for _ in range(int(rng.integers(0, 3)))]
        if not a_legs or not b_legs:
            continue
        A = wl.random_tensor(mod, a_legs, rng, fill=0.7)
        B = wl.random_tensor(mod, b_legs, rng, fill=0.7)
        same(to_device_tensor(nbk, A), to_device_tensor(nbk, B), k)
    v, w = wl.u1_leg(20, 1.0), wl.u1_leg(24, 1.0)
    A = wl.random_tensor((0,), [v, wl.flip(v)], rng)
    B = wl.random_tensor((0,), [w, wl.flip(w)], rng)
    with pytest.raises(ValueError):
        ab.compose_plan(to_device_tensor(nbk, A), to_device_tensor(nbk, B), 1)
