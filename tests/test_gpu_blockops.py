"""Parity of the data-movement / BLAS-1 block operations with the numpy calls of the reference's
NumpyBlockBackend (bit-exact for pure data movement)."""
import numpy as np
import pytest

from oracle import block_ops as ops

pytestmark = pytest.mark.gpu


def test_roundtrip_views_and_copies_bit_exact(bb, rng):
    a = rng.standard_normal((5, 6, 7, 3))
    A = bb.as_block(a)
    np.testing.assert_array_equal(bb.to_numpy(A), a)
    for perm in ([3, 0, 2, 1], [1, 0, 3, 2], [0, 1, 2, 3]):
        np.testing.assert_array_equal(bb.to_numpy(bb.permute_axes(A, perm)), a.transpose(perm))
    np.testing.assert_array_equal(bb.to_numpy(bb.reshape(A, (30, -1))), a.reshape(30, -1))
    np.testing.assert_array_equal(bb.to_numpy(bb.reshape(bb.permute_axes(A, [1, 0, 2, 3]), (30, 21))),
                                  a.transpose(1, 0, 2, 3).reshape(30, 21))
    np.testing.assert_array_equal(bb.to_numpy(A[1:4, :, 2:7:2, 1]), a[1:4, :, 2:7:2, 1])
    np.testing.assert_array_equal(bb.to_numpy(A[2]), a[2])
    np.testing.assert_array_equal(bb.to_numpy(bb.copy_block(bb.permute_axes(A, [2, 3, 0, 1]))), a.transpose(2, 3, 0, 1))
    np.testing.assert_array_equal(bb.to_numpy(bb.combine_legs(A, [[1, 2]])), ops.combine_legs(a, [[1, 2]]))
    np.testing.assert_array_equal(bb.to_numpy(bb.combine_legs(A, [[0, 1], [2, 3]], [True, False])),
                                  ops.combine_legs(a, [[0, 1], [2, 3]], [True, False]))
    c = bb.combine_legs(A, [[1, 2]], cstyles=False)
    np.testing.assert_array_equal(bb.to_numpy(bb.split_legs(c, [1], [[6, 7]], cstyles=False)), a)
    np.testing.assert_array_equal(bb.to_numpy(bb.dagger(A)), a.transpose(3, 2, 1, 0))
    with pytest.raises(ValueError):
        bb.reshape(A, (7, 7))
    with pytest.raises(ValueError):
        bb.permute_axes(A, [0, 0, 1, 2])
    with pytest.raises(IndexError):
        A[9]


def test_set_item_subblock_scatter(bb, rng):
    """`new_block[slices] = combined` of AbelianBackend::combine_legs (abelian.cpp:1212-1214)."""
    big = bb.zeros((12, 9))
    ref = np.zeros((12, 9))
    pairs = []
    for (r0, r1, c0, c1) in [(0, 5, 0, 4), (5, 12, 4, 9), (0, 5, 4, 9)]:
        blk = rng.standard_normal((r1 - r0, c1 - c0))
        ref[r0:r1, c0:c1] = blk
        pairs.append((bb.get_item(big, (slice(r0, r1), slice(c0, c1))), bb.as_block(blk)))
    bb.copy_many(pairs)
    np.testing.assert_array_equal(bb.to_numpy(big), ref)
    bb.set_item(big, (slice(2, 4), slice(1, 3)), bb.ones_block((2, 2)))
    ref[2:4, 1:3] = 1.0
    np.testing.assert_array_equal(bb.to_numpy(big), ref)


@pytest.mark.parametrize('ncol', [16, 17, 63, 64, 101, 412, 1031])
def test_row_path_alignments_and_ragged_rows(bb, rng, ncol):
    """Wave-per-row copy / mask paths: odd leading dimensions (rows start 8- but not 16-byte aligned), sub-blocks at
    odd column offsets, work items that end mid-row, source and destination misaligned differently."""
    nrow = 173
    a = rng.standard_normal((nrow, ncol + 7))
    A = bb.as_block(a)
    for c0 in (0, 1, 2, 3):
        sub = bb.get_item(A, (slice(3, nrow - 2), slice(c0, c0 + ncol)))
        np.testing.assert_array_equal(bb.to_numpy(bb.contiguous(sub)), a[3:nrow - 2, c0:c0 + ncol])
        big = bb.zeros((nrow + 1, ncol + 5))
        bb.copy_many([(bb.get_item(big, (slice(1, nrow - 4), slice(c0 + 1, c0 + 1 + ncol))), sub)])
        ref = np.zeros((nrow + 1, ncol + 5))
        ref[1:nrow - 4, c0 + 1:c0 + 1 + ncol] = a[3:nrow - 2, c0:c0 + ncol]
        np.testing.assert_array_equal(bb.to_numpy(big), ref)
    # 3-d source with a kept last axis (outer index decoded per row)
    t = rng.standard_normal((5, 7, ncol))
    T = bb.as_block(t)
    np.testing.assert_array_equal(bb.to_numpy(bb.contiguous(bb.permute_axes(T, [1, 0, 2]))), t.transpose(1, 0, 2))
    # masks along a leading axis (row gather) and along the last axis (32-bit index path), and their scatters
    m0 = rng.random(nrow) < 0.6
    m1 = rng.random(ncol + 7) < 0.6
    m0[0] = m1[0] = True
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_mask(A, m0, 0)), a[m0])
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_mask(A, m1, 1)), a[:, m1])
    np.testing.assert_array_equal(bb.to_numpy(bb.enlarge_leg(bb.as_block(a[m0]), m0, 0)), ops.enlarge_leg(a[m0], m0, 0))
    np.testing.assert_array_equal(bb.to_numpy(bb.enlarge_leg(bb.as_block(a[:, m1]), m1, 1)), ops.enlarge_leg(a[:, m1], m1, 1))
    mt = rng.random(7) < 0.6
    mt[1] = True
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_mask(T, mt, 1)), t[:, mt])


@pytest.mark.parametrize('shape,perm', [((2, 5, 37, 41, 2), (3, 4, 0, 2, 1)),      # H_eff: [p1', wR, vR, vL', p0'] -> [vL', p0', p1', vR, wR]
                                        ((5, 2, 2, 33, 29), (1, 2, 3, 4, 0)), ((29, 3, 33, 5), (2, 3, 0, 1)),
                                        ((7, 3, 40, 2), (3, 2, 1, 0)), ((4, 50, 3), (2, 1, 0)), ((3, 70, 4, 5), (1, 3, 0, 2)),
                                        ((17, 2, 19, 3), (2, 3, 0, 1)), ((2, 2, 2, 2, 2, 2, 2, 2), (7, 6, 5, 4, 3, 2, 1, 0)),
                                        ((64, 5, 65), (2, 1, 0)), ((130, 6), (1, 0)), ((6, 130), (1, 0))])
def test_permutations_with_short_inner_axes(bb, rng, shape, perm):
    """Leg rotations whose innermost axes are physical / MPO legs of extent 2-5: the tiled transposing copy flattens a
    short unit-stride axis with its contiguous neighbour (composite tile axes); bit-exact against numpy for float64 and
    complex128, plain and conjugated."""
    a = rng.standard_normal(shape)
    got = bb.to_numpy(bb.contiguous(bb.permute_axes(bb.as_block(a), list(perm))))
    np.testing.assert_array_equal(got, a.transpose(perm))
    z = a + 1j * rng.standard_normal(shape)
    Z = bb.permute_axes(bb.as_block(z), list(perm))
    np.testing.assert_array_equal(bb.to_numpy(bb.contiguous(Z)), z.transpose(perm))
    np.testing.assert_array_equal(bb.to_numpy(bb.conj(Z)), z.transpose(perm).conj())
    # and as the destination of a scatter (set_item into a permuted view of a larger block)
    big = bb.zeros(tuple(s + 1 for s in a.transpose(perm).shape))
    bb.copy_many([(bb.get_item(big, tuple(slice(1, None) for _ in shape)), bb.permute_axes(bb.as_block(a), list(perm)))])
    ref = np.zeros(tuple(s + 1 for s in a.transpose(perm).shape))
    ref[tuple(slice(1, None) for _ in shape)] = a.transpose(perm)
    np.testing.assert_array_equal(bb.to_numpy(big), ref)


def test_large_copy_spans_many_work_items(bb, rng):
    a = rng.standard_normal((3000, 701))
    A = bb.as_block(a)
    sub = bb.get_item(A, (slice(1, 2999), slice(3, 700)))
    np.testing.assert_array_equal(bb.to_numpy(bb.contiguous(sub)), a[1:2999, 3:700])
    m = rng.random(3000) < 0.5
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_mask(A, m, 0)), a[m])
    m = rng.random(701) < 0.5
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_mask(A, m, 1)), a[:, m])


def test_masks_scale_axis_fills(bb, rng):
    a = rng.standard_normal((4, 9, 5))
    A = bb.as_block(a)
    mask = rng.random(9) < 0.5
    mask[0] = True
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_mask(A, mask, 1)), ops.apply_mask(a, mask, 1))
    small = a[:, mask]
    np.testing.assert_array_equal(bb.to_numpy(bb.enlarge_leg(bb.as_block(small), mask, 1)), ops.enlarge_leg(small, mask, 1))
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_mask(A, np.zeros(9, bool), 1)), a[:, :0])
    f = rng.standard_normal(9)
    np.testing.assert_array_equal(bb.to_numpy(bb.scale_axis(A, bb.as_block(f), 1)), ops.scale_axis(a, f, 1))
    np.testing.assert_array_equal(bb.to_numpy(bb.eye_matrix(7)), np.eye(7))
    np.testing.assert_array_equal(bb.to_numpy(bb.zeros((3, 2))), np.zeros((3, 2)))
    np.testing.assert_array_equal(bb.to_numpy(bb.ones_block((3, 2))), np.ones((3, 2)))
    np.testing.assert_array_equal(bb.to_numpy(bb.eye_block([2, 3])), np.eye(6).reshape(2, 3, 2, 3))
    d = rng.standard_normal(6)
    np.testing.assert_array_equal(bb.to_numpy(bb.block_from_diagonal(bb.as_block(d))), np.diag(d))
    np.testing.assert_array_equal(bb.to_numpy(bb.get_diagonal(bb.as_block(np.diag(d)))), d)
    np.testing.assert_array_equal(bb.to_numpy(bb.tile(bb.as_block(d), 3)), np.tile(d, 3))


def test_blas1_ops(bb, rng):
    xs = [rng.standard_normal(s) for s in [(3,), (40, 50), (7, 8, 9), (1,), (200000,)]]
    ys = [rng.standard_normal(x.shape) for x in xs]
    X, Y = [bb.as_block(x) for x in xs], [bb.as_block(y) for y in ys]
    ref_norm = np.sqrt(sum(np.sum(x * x) for x in xs))
    assert abs(bb.norm_many(X) - ref_norm) <= 1e-12 * ref_norm
    ref_in = sum(np.sum(x * y) for x, y in zip(xs, ys))
    assert abs(bb.inner_many(X, Y) - ref_in) <= 1e-10 * ref_norm ** 2
    assert abs(bb.norm(X[1]) - ops.norm(xs[1])) <= 1e-12 * ops.norm(xs[1])
    assert abs(bb.inner(X[2], Y[2], True) - ops.inner(xs[2], ys[2], True)) <= 1e-11 * ops.norm(xs[2]) * ops.norm(ys[2])
    assert bb.max_abs_many(X) == max(np.abs(x).max() for x in xs)
    lc = bb.linear_combination_many(2.5, X, -0.75, Y)
    for o, x, y in zip(lc, xs, ys):
        np.testing.assert_allclose(bb.to_numpy(o), 2.5 * x - 0.75 * y, rtol=0, atol=1e-15 * 4)
    np.testing.assert_array_equal(bb.to_numpy(bb.mul(3.0, X[1])), 3.0 * xs[1])
    np.testing.assert_array_equal(bb.to_numpy(X[1] + Y[1]), xs[1] + ys[1])
    np.testing.assert_array_equal(bb.to_numpy(X[1] - Y[1]), xs[1] - ys[1])
    np.testing.assert_array_equal(bb.to_numpy(X[1] * Y[1]), xs[1] * ys[1])
    np.testing.assert_array_equal(bb.to_numpy(abs(X[2])), np.abs(xs[2]))
    np.testing.assert_allclose(bb.to_numpy(bb.sqrt(abs(X[2]))), np.sqrt(np.abs(xs[2])), rtol=1e-15)
    assert abs(bb.sum_all(X[1]) - xs[1].sum()) <= 1e-11 * np.abs(xs[1]).sum()
    assert bb.item(bb.as_block(np.array([4.25]))) == 4.25
    assert bb.get_block_element(X[1], [3, 4]) == xs[1][3, 4]
    assert bb.allclose(X[1], bb.as_block(xs[1] + 1e-12)) and not bb.allclose(X[1], Y[1])
    with pytest.raises(ValueError):
        bb.item(X[1])
    assert bb.norm_many([]) == 0.0


def test_random_normal_statistics(bb):
    r = bb.to_numpy(bb.random_normal((200000,), sigma=2.0, seed=7))
    assert abs(r.mean()) < 0.03 and abs(r.std() - 2.0) < 0.03
    r2 = bb.to_numpy(bb.random_normal((200000,), sigma=2.0, seed=7))
    np.testing.assert_array_equal(r, r2)
