"""The remaining virtuals of the reference's BlockBackend operator API (block_backend.h:243-488) against the numpy
calls NumpyBlockBackend makes for them (oracle/block_ops.py cites each call site)."""
import numpy as np
import pytest
import scipy.linalg

from oracle import block_ops as ops

pytestmark = pytest.mark.gpu


def test_comparisons_give_boolean_blocks(bb, rng):
    a, b = rng.standard_normal((7, 9)), rng.standard_normal((7, 9))
    b[2, 3] = a[2, 3]
    A, B = bb.as_block(a), bb.as_block(b)
    for got, want in ((A < B, a < b), (A <= B, a <= b), (A > B, a > b), (A >= B, a >= b), (A == B, a == b), (A != B, a != b),
                      (A > 0.25, a > 0.25), (A <= 0, a <= 0)):
        assert got.dtype == np.dtype('bool') and got.shape == a.shape
        np.testing.assert_array_equal(bb.to_numpy(got), want)
    m = A > 0.0
    assert bb.any(m) and not bb.all(m) and bb.all(A == A) and not bb.any(A != A)
    assert bb.sum_all(m) == int((a > 0).sum())
    # boolean blocks are first-class: views, copies, host round trip, use as a mask
    np.testing.assert_array_equal(bb.to_numpy(bb.permute_axes(m, [1, 0])), (a > 0).T)
    np.testing.assert_array_equal(bb.to_numpy(bb.copy_block(m[1:5, ::2])), (a > 0)[1:5, ::2])
    np.testing.assert_array_equal(bb.to_numpy(bb.as_block(a > 0)), a > 0)
    np.testing.assert_array_equal(bb.to_numpy(bb.to_dtype(m, 'float64')), (a > 0).astype(float))
    np.testing.assert_array_equal(bb.to_numpy(bb.to_dtype(A, 'bool')), a != 0)
    row = bb.as_block(a[0] > 0)
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_mask(A, row, 1)), a[:, a[0] > 0])
    assert (A == None) is False  # noqa: E711  (foreign operands fall back to identity, not to a device launch)
    with pytest.raises(ValueError):
        A < bb.as_block(a[:3])


def test_extrema_and_indices(bb, rng):
    a = rng.standard_normal((6, 5, 7))
    a[1, 2, 3] = a[4, 0, 6] = 9.5       # ties: first occurrence in C order
    a[2, 2, 2] = a[5, 4, 0] = -11.0
    A = bb.as_block(a)
    assert bb.max(A) == a.max() and bb.min(A) == a.min() and bb.max_abs(A) == np.abs(a).max()
    assert bb.abs_argmax(A) == ops.abs_argmax(a) == [2, 2, 2]
    assert bb.argmin(A) == ops.argmin(a) == [2, 2, 2]
    v = bb.permute_axes(A, [2, 0, 1])   # non-contiguous view
    assert bb.abs_argmax(v) == ops.abs_argmax(a.transpose(2, 0, 1))
    big = rng.standard_normal(300001)
    big[[77, 200000]] = 8.0
    assert bb.abs_argmax(bb.as_block(big)) == [77] and bb.argmin(bb.as_block(-big)) == [77]
    with pytest.raises(ValueError):
        bb.max(bb.zeros((0, 3)))


def test_elementwise_with_parameter(bb, rng):
    a = rng.standard_normal((40, 33))
    a[0, :4] = [0.0, -0.0, 1e-14, -1e-14]
    A = bb.as_block(a)
    np.testing.assert_array_equal(bb.to_numpy(bb.cutoff_inverse(A, 1e-10)), ops.cutoff_inverse(a, 1e-10))
    np.testing.assert_allclose(bb.to_numpy(bb.stable_log(A, 1e-3)), ops.stable_log(a, 1e-3), rtol=1e-14, atol=0)
    np.testing.assert_array_equal(bb.to_numpy(bb.angle(A)), ops.angle(a))
    p = np.abs(a) + 0.1
    np.testing.assert_allclose(bb.to_numpy(bb.as_block(p).pow(2.5)), p ** 2.5, rtol=1e-14)
    np.testing.assert_allclose(bb.to_numpy(bb.as_block(p) ** -1), p ** -1.0, rtol=1e-14)
    e = rng.uniform(-3.0, 3.0, a.shape)                                     # Block::pow(Block), numpy.cpp:265-276
    np.testing.assert_allclose(bb.to_numpy(bb.as_block(p).pow(bb.as_block(e))), p ** e, rtol=1e-13)
    np.testing.assert_allclose(bb.to_numpy(bb.permute_axes(bb.as_block(p), [1, 0]) ** bb.as_block(e.T.copy())), (p ** e).T, rtol=1e-13)
    with pytest.raises(ValueError):
        bb.as_block(p).pow(bb.as_block(e[:3]))


@pytest.mark.parametrize('shape,ax', [((5, 6, 7), 0), ((5, 6, 7), 1), ((5, 6, 7), -1), ((300, 2), 0), ((1, 9), 1), ((2000,), 0)])
def test_sum_over_axis(bb, rng, shape, ax):
    a = rng.standard_normal(shape)
    got = bb.to_numpy(bb.sum(bb.as_block(a), ax))
    np.testing.assert_allclose(got, np.sum(a, axis=ax), rtol=0, atol=1e-12 * np.abs(a).sum())
    z = a + 1j * rng.standard_normal(shape)
    np.testing.assert_allclose(bb.to_numpy(bb.sum(bb.as_block(z), ax)), np.sum(z, axis=ax), rtol=0, atol=1e-12 * np.abs(z).sum())


def test_trace_partial_and_full(bb, rng):
    a = rng.standard_normal((3, 4, 5, 4, 3, 2))
    A = bb.as_block(a)
    got = bb.to_numpy(bb.trace_partial(A, [0, 1], [4, 3], [2, 5]))
    np.testing.assert_allclose(got, ops.trace_partial(a, [0, 1], [4, 3], [2, 5]), atol=1e-13)
    got = bb.to_numpy(bb.trace_partial(A, [1], [3], [0, 2, 4, 5]))
    np.testing.assert_allclose(got, ops.trace_partial(a, [1], [3], [0, 2, 4, 5]), atol=1e-13)
    with pytest.raises(ValueError):
        bb.trace_partial(A, [0], [1], [2, 3, 4, 5])


def test_leg_permutations(bb, rng):
    a = rng.standard_normal((6, 7, 5))
    perms = [rng.permutation(6), np.arange(7), rng.permutation(5)]
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_leg_permutations(bb.as_block(a), perms)), ops.apply_leg_permutations(a, perms))

    class _Leg:
        def __init__(self, p):
            self.basis_perm, self.inverse_basis_perm = p, np.argsort(p)
    legs = [_Leg(p) for p in perms]
    fwd = bb.apply_basis_perm(bb.as_block(a), legs)
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_basis_perm(fwd, legs, inv=True)), a)
    with pytest.raises(ValueError):
        bb.apply_leg_permutations(bb.as_block(a), perms[:2])


def test_argsort_options(bb, rng):
    w = rng.standard_normal(25)
    for sort in (None, 'm<', 'm>', '<', '>', 'SM', 'LM', 'SR', 'LR'):
        np.testing.assert_array_equal(bb.argsort(bb.as_block(w), sort), ops.argsort(w, sort) if sort else np.argsort(w, kind='stable'))
    with pytest.raises(ValueError):
        bb.argsort(bb.as_block(w), 'bogus')


def test_masks_as_blocks(bb, rng):
    mask = rng.random(17) < 0.5
    mask[3] = True
    np.testing.assert_array_equal(bb.to_numpy(bb.block_from_mask(bb.as_block(mask))), ops.block_from_mask(mask))
    np.testing.assert_array_equal(bb.to_numpy(bb.block_from_mask(mask, 'complex128')), ops.block_from_mask(mask, complex))
    M = bb.as_block(mask)
    for big in range(17):
        for small in range(int(mask.sum())):
            assert bb.get_block_mask_element(M, big, small) == ops.get_block_mask_element(mask, big, small)
    assert bb.get_block_mask_element(M, 17 + 3, 5 + int(mask[:3].sum()), sum_block=5) == \
        ops.get_block_mask_element(mask, 17 + 3, 5 + int(mask[:3].sum()), sum_block=5)


@pytest.mark.parametrize('n,scale', [(1, 1.0), (6, 0.1), (40, 1.0), (97, 6.0)])
def test_matrix_exp(bb, rng, n, scale):
    a = scale * rng.standard_normal((n, n)) / np.sqrt(n)
    want = scipy.linalg.expm(a)
    got = bb.to_numpy(bb.matrix_exp(bb.as_block(a)))
    assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    h = a + a.T                      # hermitian generator: exp is symmetric positive definite
    got = bb.to_numpy(bb.matrix_exp(bb.as_block(h)))
    assert np.abs(got - scipy.linalg.expm(h)).max() <= 1e-10 * np.abs(scipy.linalg.expm(h)).max()
    z = a + 1j * scale * rng.standard_normal((n, n)) / np.sqrt(n)
    got = bb.to_numpy(bb.matrix_exp(bb.as_block(z)))
    assert np.abs(got - scipy.linalg.expm(z)).max() <= 1e-10 * np.abs(scipy.linalg.expm(z)).max()


def test_combined_index_permutations_and_outer(bb, rng):
    a = rng.standard_normal((6, 20))
    A = bb.as_block(a)
    np.testing.assert_array_equal(bb.to_numpy(bb.permute_combined_matrix(A, [2, 3], [3, 0], [4, 5], [1, 2])),
                                  ops.permute_combined_matrix(a, [2, 3], [3, 0], [4, 5], [1, 2]))
    np.testing.assert_array_equal(bb.to_numpy(bb.permute_combined_idx(A, 0, [2, 3], [1, 0])), ops.permute_combined_idx(a, 0, [2, 3], [1, 0]))
    np.testing.assert_array_equal(bb.to_numpy(bb.permute_combined_idx(A, -1, [4, 5], [1, 0])), ops.permute_combined_idx(a, -1, [4, 5], [1, 0]))
    with pytest.raises(ValueError):
        bb.permute_combined_idx(A, 2, [4, 5], [1, 0])
    x, y = rng.standard_normal((2, 3, 4)), rng.standard_normal((5, 2))
    np.testing.assert_allclose(bb.to_numpy(bb.tensor_outer(bb.as_block(x), bb.as_block(y), 1)), ops.tensor_outer(x, y, 1), atol=1e-15)


def test_random_uniform_and_misc(bb):
    u = bb.to_numpy(bb.random_uniform((400, 500), seed=7))
    assert u.min() >= -1.0 and u.max() < 1.0
    assert abs(u.mean()) < 5e-3 and abs(u.var() - 1.0 / 3.0) < 5e-3
    np.testing.assert_array_equal(u, bb.to_numpy(bb.random_uniform((400, 500), seed=7)))   # counter based: reproducible
    assert not np.array_equal(u, bb.to_numpy(bb.random_uniform((400, 500), seed=8)))
    z = bb.to_numpy(bb.random_uniform((300, 300), dtype='complex128', seed=3))
    assert z.dtype == np.complex128 and abs(np.corrcoef(z.real.ravel(), z.imag.ravel())[0, 1]) < 2e-2
    assert bb.as_scalar(np.float64(2.5)).as_float64() == 2.5 and bb.as_scalar(3, 'complex128').as_complex128() == 3 + 0j and bb.as_scalar(1.0, 'bool').as_bool() is True
    zc = bb.as_block(np.array([1.0 + 1e-18j, 2.0]))
    assert bb.real_if_close(zc, 100).dtype == np.dtype('float64')
    assert bb.real_if_close(bb.as_block(np.array([1.0 + 1e-3j])), 100).dtype == np.dtype('complex128')
    lines = bb._block_repr_lines(bb.as_block(np.arange(400.0).reshape(40, 10)), '  ', 60, 7)
    assert len(lines) == 7 and lines[3] == '  ...' and all(x.startswith('  ') for x in lines)
    assert bb.to_dtype(bb.as_block(np.ones(3)), 'float32').dtype == np.dtype('float32')   # (the six dtypes: tests/test_gpu_dtypes.py)
    with pytest.raises(NotImplementedError):
        bb.to_dtype(bb.as_block(np.ones(3)), 'float16')


def _random_tree_updates(rng, n_old=4, n_new=3):
    """Synthetic mapping data with the structure transform_tensor iterates over: every new block is tiled by
    (row slice, column slice) tree-block pairs; each pair sums a few scaled sub-blocks of old blocks whose
    multiplicities are the permuted ones."""
    J, K = 2, 2
    old_shapes, new_shapes, updates = [], [], []
    mults = [int(x) for x in rng.integers(1, 5, size=J + K)]
    perm = [int(x) for x in rng.permutation(J + K)]
    idcs1, idcs2 = perm[:2], perm[2:]
    dims1, dims2 = mults[:J], mults[J:]
    m_old, n_old_c = int(np.prod(dims1)), int(np.prod(dims2))
    pshape = [mults[i] for i in perm]
    m_new, n_new_c = int(np.prod(pshape[:2])), int(np.prod(pshape[2:]))
    for _ in range(n_old):
        old_shapes.append((m_old * int(rng.integers(1, 4)), n_old_c * int(rng.integers(1, 4))))
    for b in range(n_new):
        nr, nc = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        new_shapes.append((m_new * nr + 1, n_new_c * nc + 2))          # a margin that must stay zero
        for r in range(nr):
            for c in range(nc):
                if rng.random() < 0.25:
                    continue                                             # this tree pair gets no contribution
                terms = []
                for _ in range(int(rng.integers(1, 4))):
                    k = int(rng.integers(0, n_old))
                    r0 = m_old * int(rng.integers(0, old_shapes[k][0] // m_old))
                    c0 = n_old_c * int(rng.integers(0, old_shapes[k][1] // n_old_c))
                    terms.append((float(rng.standard_normal()), k, (r0, r0 + m_old), (c0, c0 + n_old_c)))
                updates.append((b, (m_new * r, m_new * (r + 1)), (n_new_c * c, n_new_c * (c + 1)), dims1, idcs1, dims2, idcs2, terms))
    return old_shapes, new_shapes, updates


def test_tree_block_updates_in_one_launch(bb, rng):
    """Row f.4: the block arithmetic of TreePairMapping::transform_tensor (zeros, get_item * coeff summed over terms,
    permute_combined_matrix, set_item) as one strided linear-combination launch, against the oracle's call-by-call
    restatement, on synthetic mapping data (the fusion-tree layer that produces real mappings is host bookkeeping
    outside this repo: parity on SU(2) data is unpinned)."""
    for _ in range(12):
        old_shapes, new_shapes, updates = _random_tree_updates(rng)
        old = [rng.standard_normal(sh) for sh in old_shapes]
        want = ops.transform_blocks(old, new_shapes, updates)
        got = bb.transform_blocks([bb.as_block(a) for a in old], new_shapes, updates)
        for g, w in zip(got, want):
            np.testing.assert_allclose(bb.to_numpy(g), w, rtol=0, atol=1e-13 * max(1.0, np.abs(w).max()))
    # the primitive itself: accumulate, permuted sources, scalar (0-d) and empty views, shape check
    a, b = rng.standard_normal((5, 6, 7)), rng.standard_normal((7, 5, 6))
    out = rng.standard_normal((5, 6, 7))
    O = bb.as_block(out)
    bb.lincomb_many([(O, [(2.0, bb.as_block(a)), (-0.5, bb.permute_axes(bb.as_block(b), [1, 2, 0]))], True)])
    np.testing.assert_allclose(bb.to_numpy(O), out + 2.0 * a - 0.5 * b.transpose(1, 2, 0), atol=1e-14)
    Z = bb.as_block(out)
    bb.lincomb_many([(bb.get_item(Z, (slice(1, 3), slice(None), slice(0, 7, 2))), [], False)])     # no terms: zero the view
    ref = out.copy()
    ref[1:3, :, 0:7:2] = 0.0
    np.testing.assert_array_equal(bb.to_numpy(Z), ref)
    with pytest.raises(ValueError):
        bb.lincomb_many([(O, [(1.0, bb.as_block(b))], False)])


def test_vector_norms_of_every_order(bb, rng):
    """``norm(a, order)`` = ``np.linalg.norm(a.ravel(), ord=order)`` (numpy.cpp:898-913) for the orders numpy defines on
    vectors, on real, complex and non-contiguous blocks."""
    x = rng.standard_normal((13, 7, 5))
    x[2, 3, :] = 0.0
    z = rng.standard_normal((9, 11)) + 1j * rng.standard_normal((9, 11))
    for arr in (x, z):
        blk = bb.as_block(arr)
        view = bb.permute_axes(blk, list(range(arr.ndim))[::-1])
        for order in (2, None, 1, np.inf, -np.inf, 0, 3, 0.5, 7.5):
            want = np.linalg.norm(arr.ravel(), ord=order)
            for b in (blk, view):
                got = bb.norm(b) if order is None else bb.norm(b, order)
                assert abs(got - want) <= 1e-12 * max(1.0, abs(want)), (order, got, want)
    assert bb.norm(bb.as_block(np.zeros((0, 4))), 1) == 0.0
    # per-axis form (numpy.cpp:904): a block, for the sum-type orders
    for arr in (x, z):
        for ax in range(arr.ndim):
            for order in (2, None, 1, 0, 3, 0.5):
                want = np.linalg.norm(arr, ord=2 if order is None else order, axis=ax)
                got = bb.to_numpy(bb.norm(bb.as_block(arr), order, axis=ax))
                assert got.shape == want.shape and np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (order, ax)
    with pytest.raises(NotImplementedError):
        bb.norm(bb.as_block(x), np.inf, axis=0)


def test_get_item_with_negative_steps(bb, rng):
    """numpy's basic indexing with negative slice steps (AxisSlice of block_backend.h:42-46), alone, combined with
    ints / positive slices / an index array, on real, complex and permuted blocks."""
    x = rng.standard_normal((6, 9, 5))
    z = rng.standard_normal((7, 8)) + 1j * rng.standard_normal((7, 8))
    X, Z = bb.as_block(x), bb.as_block(z)
    for key in ((slice(None, None, -1),), (slice(4, 0, -2), slice(None), slice(None, None, -1)), (2, slice(None, None, -3), slice(1, 4)),
                (slice(None), slice(7, 1, -1), 0), (slice(1, 1, -1),), (slice(None, None, -1), [0, 3, 3, 8])):
        np.testing.assert_array_equal(bb.to_numpy(bb.get_item(X, key)), x[key])
    np.testing.assert_array_equal(bb.to_numpy(bb.get_item(Z, (slice(None, None, -1), slice(6, 2, -2)))), z[::-1, 6:2:-2])
    m = x > 0.3                                                           # boolean blocks: index arrays and reversed slices
    M = bb.as_block(m)
    np.testing.assert_array_equal(bb.to_numpy(bb.get_item(M, (slice(None, None, -1), [1, 0, 7]))), m[::-1, [1, 0, 7]])
    assert bb.get_item(M, (slice(None, None, -1),)).is_bool
    Xt = bb.permute_axes(X, [2, 0, 1])
    np.testing.assert_array_equal(bb.to_numpy(bb.get_item(Xt, (slice(None, None, -1), slice(None), slice(None, None, -2)))),
                                  x.transpose(2, 0, 1)[::-1, :, ::-2])


def test_boolean_blocks_are_refused_by_the_numeric_kernels(bb, rng):
    """Boolean blocks have 1-byte storage; the float64 / complex128 kernels must refuse them (TypeError or
    NotImplementedError) instead of reading 8-byte words past the buffer.  to_dtype converts explicitly."""
    m = bb.as_block(rng.standard_normal((6, 6)) > 0)
    x = bb.as_block(rng.standard_normal((6, 6)))
    for call in (lambda: bb.norm(m), lambda: bb.inner(m, m, True), lambda: bb.matrix_dot(m, x), lambda: bb.matrix_svd(m),
                 lambda: bb.scale_axis(m, bb.as_block(np.ones(6)), 0), lambda: bb.linear_combination(1.0, m, 1.0, m), lambda: bb.sqrt(m),
                 lambda: bb.eigh(m), lambda: bb.matrix_qr(m, False)):
        with pytest.raises((TypeError, NotImplementedError)):
            call()
    f = bb.to_dtype(m, 'float64')
    assert abs(bb.norm(f) - np.linalg.norm(bb.to_numpy(m).astype(float))) < 1e-14
