"""The reference's one-block-at-a-time call pattern served by ONE grouped launch (cyten_amd/deferred.py):
replay of the contraction loop of abelian_compose_worker (abelian.cpp:1424-1460) through
DeferredBlockBackend, checked against the oracle and against the number of launches."""
import numpy as np
import pytest

from cyten_amd import abelian as ab
from cyten_amd import workloads as wl
from oracle import abelian_ref as ref

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope='module')
def dbb():
    from cyten_amd.deferred import DeferredBlockBackend
    return DeferredBlockBackend('cuda:0')


def test_reference_loop_becomes_one_launch(dbb):
    A, B = wl.config_u1_mps(256)
    a, b = ab.AbelianTensor.from_spec(dbb, A), ab.AbelianTensor.from_spec(dbb, B)
    plan = ab.compose_plan(a, b, 1)
    a2, b2 = ab._compose_operands(dbb, a, b, 1, plan)
    dbb.flush()
    f0, d0 = dbb.n_flushes, dbb.n_deferred
    res_blocks = []
    for pairs, shp in zip(plan.pairs, plan.res_shapes):     # the reference's loop, call for call
        i, j = pairs[0]
        block = dbb.matrix_dot(a2[i], b2[j])
        for i, j in pairs[1:]:
            block = block + dbb.matrix_dot(a2[i], b2[j])   # Block::operator+
        res_blocks.append(dbb.reshape(block, shp))
    n_dot = sum(len(p) for p in plan.pairs)
    assert dbb.n_deferred - d0 == n_dot and dbb.n_flushes == f0          # nothing launched yet
    assert all(tuple(blk.shape) == tuple(shp) for blk, shp in zip(res_blocks, plan.res_shapes))
    first = dbb.to_numpy(res_blocks[0])                                   # first observation: ONE launch for all
    assert dbb.n_flushes == f0 + 1
    blocks, bi, _ = ref.compose(A, B, 1)
    np.testing.assert_array_equal(plan.res_block_inds, bi)
    scale = max(np.abs(x).max() for x in blocks)
    assert np.abs(first - blocks[0]).max() <= TOL * scale
    for got, want in zip(res_blocks, blocks):
        assert np.abs(dbb.to_numpy(got) - want).max() <= TOL * scale
    assert dbb.n_flushes == f0 + 1                                        # and no further launch


def test_lazy_blocks_feed_other_kernels_and_chains(dbb, rng):
    a, b, c = (rng.standard_normal(s) for s in [(30, 20), (20, 25), (25, 10)])
    x = dbb.matrix_dot(dbb.as_block(a), dbb.as_block(b))             # pending
    y = dbb.matrix_dot(x, dbb.as_block(c))                           # depends on a pending product
    z = dbb.permute_axes(y, [1, 0])                                  # still metadata
    assert abs(dbb.norm(z) - np.linalg.norm(a @ b @ c)) <= TOL * np.linalg.norm(a @ b @ c)
    np.testing.assert_allclose(dbb.to_numpy(z), (a @ b @ c).T, atol=1e-11)
    u, s_, vh = dbb.matrix_svd(dbb.matrix_dot(dbb.as_block(a), dbb.as_block(b)))   # a decomposition observes
    np.testing.assert_allclose(dbb.to_numpy(s_), np.linalg.svd(a @ b, compute_uv=False), atol=1e-11)
    w = dbb.matrix_dot(dbb.as_block(a), dbb.as_block(b)) + dbb.as_block(np.ones((30, 25)))  # lazy + real block
    np.testing.assert_allclose(dbb.to_numpy(w), a @ b + 1.0, atol=1e-12)
    with pytest.raises(ValueError):
        dbb.matrix_dot(dbb.as_block(a), dbb.as_block(c))


def test_per_sector_svd_loop_becomes_one_batched_call(dbb):
    """AbelianBackend::svd calls bb.matrix_svd once per coupled-charge block (abelian.cpp:3499-3541); the
    first thing that looks at a result is truncate_singular_values reading S (abelian.cpp:3631)."""
    A, B = wl.config_u1_mps(256)
    oracle = ref.theta_tdot_svd(A, B, chi_max=100)
    a, b = ab.AbelianTensor.from_spec(dbb, A), ab.AbelianTensor.from_spec(dbb, B)
    theta = ab.compose(dbb, a, b, 1)
    mv = ab.combine_legs_to_matrix(dbb, theta, 2)
    dbb.flush()
    n0 = dbb.n_decomp_batches
    usv = [dbb.matrix_svd(blk) for blk in mv.blocks]          # the reference's per-sector loop
    assert dbb.n_decomp_batches == n0                          # nothing ran yet
    assert all(u.shape == (blk.shape[0], min(blk.shape)) for (u, _, _), blk in zip(usv, mv.blocks))
    S = [dbb.to_numpy(s) for _, s, _ in usv]                   # first observation
    assert dbb.n_decomp_batches == n0 + 1                      # ONE batched call for all sectors
    for s, (_, sref, _) in zip(S, oracle['usv']):
        assert np.abs(s - sref).max() <= TOL * sref[0]
    for (u, s, vh), m in zip(usv, oracle['matrices']):
        rec = (dbb.to_numpy(u) * dbb.to_numpy(s)) @ dbb.to_numpy(vh)
        assert np.abs(rec - m).max() <= TOL * np.abs(m).max() * max(m.shape)
    q, r = dbb.matrix_qr(mv.blocks[0], False)
    w, v = dbb.eigh(dbb.matrix_dot(mv.blocks[0], dbb.permute_axes(mv.blocks[0], [1, 0])))
    m0 = oracle['matrices'][0]
    np.testing.assert_allclose(dbb.to_numpy(q) @ dbb.to_numpy(r), m0, atol=1e-10 * np.abs(m0).max() * max(m0.shape))
    np.testing.assert_allclose(dbb.to_numpy(w), np.linalg.eigvalsh(m0 @ m0.T), atol=1e-9 * np.abs(m0).max() ** 2 * max(m0.shape))


def test_complex_decompositions_run_at_once(dbb, rng):
    """Lazy decomposition outputs are typed float64; complex blocks (small and beyond the in-LDS limit) are decomposed
    eagerly after the pending products have been flushed -- also when the block is itself a pending product."""
    a = rng.standard_normal((90, 70)) + 1j * rng.standard_normal((90, 70))
    b = rng.standard_normal((70, 80)) + 1j * rng.standard_normal((70, 80))
    prod = dbb.matrix_dot(dbb.as_block(a), dbb.as_block(b))
    u, s, vh = dbb.matrix_svd(prod)
    u, s, vh = dbb.to_numpy(u), dbb.to_numpy(s), dbb.to_numpy(vh)
    ref = a @ b
    assert np.abs((u * s) @ vh - ref).max() <= 1e-10 * np.linalg.norm(ref)
    assert np.abs(u.conj().T @ u - np.eye(80)).max() <= 1e-10
    q, r = dbb.matrix_qr(dbb.as_block(a), False)
    assert np.abs(dbb.to_numpy(q) @ dbb.to_numpy(r) - a).max() <= 1e-10 * np.linalg.norm(a)
    h = a[:70] + a[:70].conj().T
    w, v = dbb.eigh(dbb.as_block(h))
    assert np.abs(dbb.to_numpy(w) - np.linalg.eigvalsh(h)).max() <= 1e-10 * np.linalg.norm(h)


def test_product_decomposition_product_chain_and_replaced_nodes(dbb, rng):
    """(a) matrix_dot -> matrix_qr of the lazy product -> matrix_dot with a lazy Q, observed only at the end,
    runs in dependency order (product, decomposition, product) without a nested flush; (b) a lazy block stays readable
    after it was reshaped / permuted / summed into a longer chain."""
    a, b, c = (rng.standard_normal(s) for s in [(40, 30), (30, 35), (20, 40)])
    A, B, Cm = dbb.as_block(a), dbb.as_block(b), dbb.as_block(c)
    dbb.flush()
    f0, d0 = dbb.n_flushes, dbb.n_decomp_batches
    x = dbb.matrix_dot(A, B)                       # pending product
    q, r = dbb.matrix_qr(x, False)                 # pending decomposition of a pending product
    u, s, vh = dbb.matrix_svd(r)                   # ... of a pending decomposition output
    y = dbb.matrix_dot(Cm, q)                      # product with a lazy decomposition output as operand
    assert dbb.n_flushes == f0 and dbb.n_decomp_batches == d0
    got = dbb.to_numpy(y)                          # first observation
    assert dbb.n_flushes == f0 + 2 and dbb.n_decomp_batches == d0 + 2   # product, qr, svd, product
    qn, rn = dbb.to_numpy(q), dbb.to_numpy(r)
    np.testing.assert_allclose(qn @ rn, a @ b, atol=1e-11 * np.abs(a @ b).max() * 40)
    np.testing.assert_allclose(got, c @ qn, atol=1e-12 * 40)
    np.testing.assert_allclose(dbb.to_numpy(s), np.linalg.svd(a @ b, compute_uv=False), atol=1e-10 * np.linalg.norm(a @ b))
    # (b) the original node after metadata operations on it
    p = dbb.matrix_dot(A, B)
    p2 = dbb.reshape(p, (40, 5, 7))
    p3 = dbb.permute_axes(p2, [2, 0, 1])
    np.testing.assert_allclose(dbb.to_numpy(p), a @ b, atol=1e-12 * 30)          # read the ORIGINAL first
    np.testing.assert_allclose(dbb.to_numpy(p3), (a @ b).reshape(40, 5, 7).transpose(2, 0, 1), atol=1e-12 * 30)
    np.testing.assert_allclose(dbb.to_numpy(p2), (a @ b).reshape(40, 5, 7), atol=1e-12 * 30)
    t1, t2 = dbb.matrix_dot(A, B), dbb.matrix_dot(A, B)
    tot = t1 + t2                                   # folded into one two-segment product
    np.testing.assert_allclose(dbb.to_numpy(tot), 2 * (a @ b), atol=1e-12 * 60)
    np.testing.assert_allclose(dbb.to_numpy(t1), a @ b, atol=1e-12 * 30)         # the addend is still materialisable


def test_folded_addend_used_as_an_operand_and_failed_launches_keep_the_queue(dbb, rng):
    """Two holes of flush() the round-2 advisor named: (1) an addend that `+` folded into a longer chain left the queue; used
    as an operand of another pending product it was never produced and flush raised a spurious 'cyclic dependency';
    (2) a launch that raised inside flush emptied the queues, and the surviving lazy blocks read None afterwards."""
    a, b, c, d, e = (rng.standard_normal((6, 6)) for _ in range(5))
    A, B, C_, D, E = (dbb.as_block(x) for x in (a, b, c, d, e))
    p1, p2 = dbb.matrix_dot(A, B), dbb.matrix_dot(C_, D)
    s = p1 + p2                                    # folds p1 and p2: both leave the queue
    q = dbb.matrix_dot(p1, E)                      # ... but p1 is an operand of a new pending product
    v = dbb.matrix_dot(dbb.permute_axes(p2, [1, 0]), E)   # and a VIEW of the other addend of another
    np.testing.assert_allclose(dbb.to_numpy(q), (a @ b) @ e, rtol=0, atol=1e-12)
    np.testing.assert_allclose(dbb.to_numpy(v), (c @ d).T @ e, rtol=0, atol=1e-12)
    np.testing.assert_allclose(dbb.to_numpy(s), a @ b + c @ d, rtol=0, atol=1e-12)
    # (2): a decomposition whose batched call raises in the same flush as a healthy product (the launch is made to fail here:
    #      what matters is what the queue looks like afterwards)
    from cyten_amd._lib import LinAlgError, CYB_ERR_NOCONV
    x = dbb.as_block(rng.standard_normal((20, 20)))
    u, s_, vh = dbb.matrix_svd(x)
    good = dbb.matrix_dot(A, B)
    real_run = dbb._run_decomps
    calls = []

    def failing(nodes):
        calls.append(len(nodes))
        raise LinAlgError(CYB_ERR_NOCONV, 'injected failure')

    dbb._run_decomps = failing
    try:
        with pytest.raises(LinAlgError):
            dbb.to_numpy(s_)
        np.testing.assert_allclose(dbb.to_numpy(good), a @ b, rtol=0, atol=1e-12)   # the product of that flush is there
        with pytest.raises(LinAlgError):
            dbb.to_numpy(u)                                                            # the SAME error again, not an AttributeError on None
        assert calls == [1, 1]
    finally:
        dbb._run_decomps = real_run
    un, sn, vn = dbb.to_numpy(u), dbb.to_numpy(s_), dbb.to_numpy(vh)                   # the node stayed on the queue: it runs now
    np.testing.assert_allclose((un * sn) @ vn, dbb.to_numpy(x), rtol=0, atol=1e-12)
