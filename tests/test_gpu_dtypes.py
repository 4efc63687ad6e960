"""The six dtypes of the reference (include/cyten/block_backend/dtypes.h:12-21: bool, int64, float32, complex64, float64,
complex128) at block level, as the reference's tensor tests use them (tests/python_tests/test_tensors.py:212-218, :570-577,
:613-620, :684-692, :828-836: as_dtype(complex64) / to_backend(dtype), DiagonalTensor(dtype=float32 | bool), dtype
equality, almost_equal of the converted tensor).  float32 / complex64 / int64 blocks are held in double words on the device
and rounded to the nominal type on store; dtypes and promotion follow numpy (numpy.cpp:1131-1138 is np.asarray(a, dtype))."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL = ['bool', 'int64', 'float32', 'complex64', 'float64', 'complex128']


def test_to_dtype_accepts_the_six_dtypes_and_matches_numpy(bb, rng):
    x = rng.standard_normal((5, 7)) * 10
    z = x + 1j * rng.standard_normal((5, 7))
    m = rng.random((5, 7)) < 0.5
    for src in (x, z, m):
        blk = bb.as_block(src)
        for name in ALL:
            if np.iscomplexobj(src) and name in ('int64',):
                continue                                                   # (numpy refuses complex -> int without a warning path)
            got = bb.to_dtype(blk, name)
            assert got.dtype == np.dtype(name) and bb.get_dtype(got) == np.dtype(name)
            out = bb.to_numpy(got)
            with np.errstate(all='ignore'):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    want = src.astype(name)
            assert out.dtype == np.dtype(name) and np.array_equal(out, want)
            bb.test_block_sanity(got, expect_shape=(5, 7), expect_dtype=name)
    # round trips: widening is exact, narrowing rounds once
    f32 = bb.to_dtype(bb.as_block(x), 'float32')
    assert np.array_equal(bb.to_numpy(bb.to_dtype(f32, 'float64')), x.astype(np.float32).astype(np.float64))
    assert bb.to_dtype(f32, 'float64').dtype == np.dtype('float64')
    assert bb.to_dtype(f32, 'complex64').dtype == np.dtype('complex64')
    with pytest.raises(NotImplementedError):
        bb.to_dtype(bb.as_block(x), 'float16')


def test_blocks_from_single_precision_arrays_keep_their_dtype(bb, rng):
    a32 = rng.standard_normal((4, 3)).astype(np.float32)
    c64 = (rng.standard_normal((4, 3)) + 1j * rng.standard_normal((4, 3))).astype(np.complex64)
    i64 = rng.integers(-5, 5, (4, 3))
    for arr in (a32, c64, i64):
        blk = bb.as_block(arr)
        assert blk.dtype == arr.dtype and np.array_equal(bb.to_numpy(blk), arr)
    assert bb.as_block(a32, dtype='float64').dtype == np.dtype('float64')
    assert bb.as_block(bb.as_block(a32), dtype='complex64').dtype == np.dtype('complex64')


def test_promotion_follows_numpy(bb, rng):
    a = rng.standard_normal((6, 6))
    b = rng.standard_normal((6, 6))
    c = a + 1j * b
    f32a, f32b = bb.to_dtype(bb.as_block(a), 'float32'), bb.to_dtype(bb.as_block(b), 'float32')
    c64 = bb.to_dtype(bb.as_block(c), 'complex64')
    i64 = bb.to_dtype(bb.as_block(np.round(a * 4)), 'int64')
    f64 = bb.as_block(a)
    na, nb, nc, ni = a.astype(np.float32), b.astype(np.float32), c.astype(np.complex64), np.round(a * 4).astype(np.int64)
    cases = [(f32a + f32b, na + nb), (f32a * f32b, na * nb), (f32a - f32b, na - nb), (f32a / f32b, na / nb),
             (f32a + f64, na + a), (c64 * f32a, nc * na), (c64 + c64, nc + nc), (i64 + i64, ni + ni), (i64 * i64, ni * ni),
             (i64 - i64, ni - ni), (i64 * f64, ni * a), (i64 + f32a, ni + na), (bb.abs(c64), np.abs(nc)), (bb.abs(i64), np.abs(ni)),
             (bb.sqrt(bb.abs(f32a)), np.sqrt(np.abs(na))), (bb.real(c64), nc.real), (bb.conj(c64), nc.conj()),
             (bb.mul(2.5, f32a), np.float32(2.5) * na), (bb.linear_combination(0.5, f32a, 2.0, f32b), np.float32(0.5) * na + np.float32(2.0) * nb)]
    for got, want in cases:
        assert got.dtype == want.dtype, (got.dtype, want.dtype)
        out = bb.to_numpy(got)
        assert out.dtype == want.dtype
        tol = 0 if want.dtype.kind == 'i' else (4e-7 if want.dtype.itemsize in (4, 8) and want.dtype.kind in 'fc' and want.dtype != np.float64 else 1e-14)
        assert np.abs(out - want).max() <= tol * max(1.0, np.abs(want).max())
    # comparisons give bool blocks whatever the operand dtype
    assert (f32a < f32b).dtype == np.dtype('bool') and np.array_equal(bb.to_numpy(f32a < f32b), na < nb)


def test_views_of_single_precision_blocks_keep_dtype_and_memory(bb, rng):
    c = (rng.standard_normal((4, 5, 6)) + 1j * rng.standard_normal((4, 5, 6))).astype(np.complex64)
    blk = bb.as_block(c)
    p = bb.permute_axes(blk, [2, 0, 1])
    r = bb.reshape(blk, (20, 6))
    g = bb.get_item(blk, (slice(1, 3), slice(None), 2))
    for v, want in ((p, np.transpose(c, [2, 0, 1])), (r, c.reshape(20, 6)), (g, c[1:3, :, 2])):
        assert v.dtype == np.dtype('complex64') and v.buf is blk.buf          # metadata only, as for float64 blocks
        assert np.array_equal(bb.to_numpy(v), want)
    cp = bb.copy_block(p)
    assert cp.dtype == np.dtype('complex64') and np.array_equal(bb.to_numpy(cp), np.transpose(c, [2, 0, 1]))


def test_creation_functions_take_the_six_dtypes(bb):
    for name in ('float32', 'complex64', 'int64', 'float64', 'complex128'):
        z = bb.zeros((3, 4), dtype=name)
        assert z.dtype == np.dtype(name) and np.array_equal(bb.to_numpy(z), np.zeros((3, 4), dtype=name))
        e = bb.eye_matrix(4, dtype=name)
        assert e.dtype == np.dtype(name) and np.array_equal(bb.to_numpy(e), np.eye(4, dtype=name))
        o = bb.ones_block((2, 3), dtype=name)
        assert o.dtype == np.dtype(name) and np.array_equal(bb.to_numpy(o), np.ones((2, 3), dtype=name))
    r = bb.random_normal((64, 64), dtype='float32', seed=3)
    out = bb.to_numpy(r)
    assert r.dtype == np.dtype('float32') and out.dtype == np.float32 and 0.8 < out.std() < 1.2
    s = bb.as_scalar(1.0 / 3.0, 'float32')
    assert s.dtype == np.dtype('float32') and s.as_float64() == float(np.float32(1.0 / 3.0))
    d = bb.block_from_mask(np.array([True, False, True]), dtype='float32')       # test_tensors.py:684-692: diagonal of a mask as float32
    assert d.dtype == np.dtype('float32')


def test_hot_path_in_single_precision(bb, rng):
    """products and decompositions of float32 / complex64 blocks: computed in double precision, stored in the input type"""
    a = rng.standard_normal((40, 30)).astype(np.float32)
    b = rng.standard_normal((30, 50)).astype(np.float32)
    prod = bb.matrix_dot(bb.as_block(a), bb.as_block(b))
    assert prod.dtype == np.dtype('float32')
    assert np.abs(bb.to_numpy(prod) - a.astype(np.float64) @ b.astype(np.float64)).max() <= 2e-6 * np.abs(a @ b).max()
    c = (rng.standard_normal((30, 20)) + 1j * rng.standard_normal((30, 20))).astype(np.complex64)
    u, s, vh = bb.matrix_svd(bb.as_block(c))
    assert u.dtype == np.dtype('complex64') and s.dtype == np.dtype('float32') and vh.dtype == np.dtype('complex64')
    un, sn, vn = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
    assert np.abs((un * sn) @ vn - c).max() <= 1e-5 * np.linalg.norm(c)
    assert np.abs(sn - np.linalg.svd(c.astype(np.complex128), compute_uv=False)).max() <= 1e-6 * np.linalg.norm(c)
    q, r = bb.matrix_qr(bb.as_block(a), False)
    assert q.dtype == np.dtype('float32') and np.abs(bb.to_numpy(q) @ bb.to_numpy(r) - a).max() <= 1e-5 * np.linalg.norm(a)
    h = (a[:30, :30] + a[:30, :30].T).astype(np.float32)
    w, v = bb.eigh(bb.as_block(h))
    assert w.dtype == np.dtype('float32') and v.dtype == np.dtype('float32')
    assert np.abs(bb.to_numpy(w) - np.linalg.eigvalsh(h.astype(np.float64))).max() <= 1e-5 * np.linalg.norm(h)
    # mixed with float64: the product is float64 (numpy's promotion), nothing is rounded
    mixed = bb.matrix_dot(bb.as_block(a), bb.as_block(b.astype(np.float64)))
    assert mixed.dtype == np.dtype('float64')
    assert np.abs(bb.to_numpy(mixed) - a.astype(np.float64) @ b.astype(np.float64)).max() <= 1e-12 * np.abs(a @ b).max()


def test_assignment_into_a_single_precision_block_rounds_like_numpy(bb, rng):
    x = rng.standard_normal((4, 4)).astype(np.float32)
    y = rng.standard_normal((2, 4))
    blk = bb.as_block(x)
    bb.set_item(blk, (slice(0, 2), slice(None)), bb.as_block(y))
    want = x.copy()
    want[0:2] = y
    assert blk.dtype == np.dtype('float32') and np.array_equal(bb.to_numpy(blk), want)


def test_float64_path_is_untouched_by_the_policy(bb, rng):
    """no tagged block in the call: same objects, no rounding, float64 in and out (the headline path pays one attribute read)"""
    a = rng.standard_normal((8, 8))
    x = bb.as_block(a)
    y = bb.linear_combination(1.0, x, 1.0, x)
    assert y.dtype == np.dtype('float64') and np.array_equal(bb.to_numpy(y), a + a)
    third = bb.mul(1.0 / 3.0, x)
    assert np.array_equal(bb.to_numpy(third), a * (1.0 / 3.0))                   # not rounded to float32
