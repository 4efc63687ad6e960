"""complex128 blocks on the tdot path (the reference's second dtype, numpy.cpp dispatches every virtual on
it): storage / data movement / BLAS-1 / grouped GEMM through the real f64-MFMA kernel, against numpy.
Decompositions of complex blocks are not on the device path yet and must say so."""
import numpy as np
import pytest

from cyten_amd import abelian as ab
from cyten_amd import workloads as wl

pytestmark = pytest.mark.gpu
TOL = 1e-10


def crandn(rng, shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def test_complex_roundtrip_views_conj(bb, rng):
    a = crandn(rng, (5, 7, 3))
    x = bb.as_block(a)
    assert x.dtype == np.complex128 and not bb.is_real(x)
    np.testing.assert_array_equal(bb.to_numpy(x), a)
    np.testing.assert_array_equal(bb.to_numpy(bb.permute_axes(x, [2, 0, 1])), a.transpose(2, 0, 1))
    np.testing.assert_array_equal(bb.to_numpy(bb.reshape(x, (35, 3))), a.reshape(35, 3))
    np.testing.assert_array_equal(bb.to_numpy(bb.get_item(x, (slice(1, 4), 2, slice(None)))), a[1:4, 2, :])
    np.testing.assert_array_equal(bb.to_numpy(bb.conj(x)), a.conj())
    np.testing.assert_array_equal(bb.to_numpy(bb.dagger(x)), a.conj().transpose(2, 1, 0))
    np.testing.assert_array_equal(bb.to_numpy(bb.real(x)), a.real)
    np.testing.assert_array_equal(bb.to_numpy(bb.imag(x)), a.imag)
    np.testing.assert_array_equal(bb.to_numpy(bb.as_complex(bb.as_block(a.real))), a.real.astype(complex))
    z = bb.zeros((3, 2), dtype=np.complex128)
    assert z.dtype == np.complex128 and not bb.to_numpy(z).any()
    m = rng.random(7) < 0.5
    np.testing.assert_array_equal(bb.to_numpy(bb.apply_mask(x, m, 1)), a[:, m, :])
    f = rng.standard_normal(7)
    np.testing.assert_allclose(bb.to_numpy(bb.scale_axis(x, bb.as_block(f), 1)), a * f[None, :, None], atol=1e-15)


def test_complex_blas1(bb, rng):
    blocks = [crandn(rng, s) for s in [(4, 5), (1,), (33, 17), (0, 3)]]
    others = [crandn(rng, b.shape) for b in blocks]
    X, Y = [bb.as_block(b) for b in blocks], [bb.as_block(b) for b in others]
    n2 = np.sqrt(sum(np.sum(np.abs(b) ** 2) for b in blocks))
    assert abs(bb.norm_many(X) - n2) <= TOL * n2
    ip = sum(np.vdot(b, c) for b, c in zip(blocks, others))
    got = bb.inner_many(X, Y)
    assert isinstance(got, complex) and abs(got - ip) <= TOL * abs(ip)
    a_c, b_c = 0.7 - 0.2j, -1.3 + 0.5j
    for got, b, c in zip(bb.linear_combination_many(a_c, X, b_c, Y), blocks, others):
        np.testing.assert_allclose(bb.to_numpy(got), a_c * b + b_c * c, atol=1e-13)
    for got, b in zip(bb.mul_many(2.0, X), blocks):
        np.testing.assert_allclose(bb.to_numpy(got), 2.0 * b, atol=1e-14)
    # mixed real / complex
    r = rng.standard_normal((4, 5))
    np.testing.assert_allclose(bb.to_numpy(bb.as_block(r) + X[0]), r + blocks[0], atol=1e-14)
    np.testing.assert_allclose(bb.to_numpy(X[0] - Y[0]), blocks[0] - others[0], atol=1e-14)


@pytest.mark.parametrize('shape', [(1, 1, 1), (5, 3, 7), (64, 64, 64), (130, 90, 77), (33, 300, 29)])
def test_complex_matrix_dot(bb, rng, shape):
    M, N, K = shape
    a, b = crandn(rng, (M, K)), crandn(rng, (K, N))
    c = bb.to_numpy(bb.matrix_dot(bb.as_block(a), bb.as_block(b)))
    ref = a @ b
    assert c.dtype == np.complex128 and np.abs(c - ref).max() <= TOL * max(1.0, np.abs(ref).max())


def test_complex_grouped_gemm_ksplit_and_views(bb, rng):
    groups_np = []
    for _ in range(12):
        M, N = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        groups_np.append([(crandn(rng, (M, k)), crandn(rng, (k, N))) for k in rng.integers(1, 60, size=int(rng.integers(1, 4)))])
    # one mixed group (real x complex) and transposed operand views
    groups_np.append([(rng.standard_normal((20, 30)).astype(complex).real, crandn(rng, (30, 11)))])
    groups = [[(bb.as_block(a), bb.as_block(b)) for a, b in g] for g in groups_np]
    at = crandn(rng, (40, 25))
    bt = crandn(rng, (35, 40))
    groups.append([(bb.permute_axes(bb.as_block(at), [1, 0]), bb.permute_axes(bb.as_block(bt), [1, 0]))])
    groups_np.append([(at.T, bt.T)])
    outs = bb.matrix_dot_grouped(groups)
    for o, g in zip(outs, groups_np):
        ref = sum(a @ b for a, b in g)
        assert np.abs(bb.to_numpy(o) - ref).max() <= TOL * max(1.0, np.abs(ref).max())


def test_complex_abelian_compose_and_norm(bb, rng):
    """U(1) theta = A.B with complex blocks: same sector matching, complex GEMM, vs the dense contraction."""
    A, B = wl.config_u1_mps(64)
    for t in (A, B):
        t.blocks = [b + 1j * rng.standard_normal(b.shape) for b in t.blocks]
    a, b = ab.AbelianTensor.from_spec(bb, A), ab.AbelianTensor.from_spec(bb, B)
    theta = ab.compose(bb, a, b, 1)
    dense = np.tensordot(a.to_dense(bb), b.to_dense(bb), axes=([2], [0]))
    got = theta.to_dense(bb)
    assert got.dtype == np.complex128 and np.abs(got - dense).max() <= TOL * np.abs(dense).max()
    assert abs(ab.norm(bb, theta) - np.linalg.norm(dense)) <= TOL * np.linalg.norm(dense)
    ip = ab.inner(bb, theta, theta)
    assert abs(ip - np.linalg.norm(dense) ** 2) <= TOL * np.linalg.norm(dense) ** 2


def test_complex_decompositions_say_not_implemented(bb, rng):
    x = bb.as_block(crandn(rng, (6, 6)))
    for call in (lambda: bb.matrix_svd(x), lambda: bb.matrix_qr(x, False), lambda: bb.eigh(x)):
        with pytest.raises(NotImplementedError):
            call()


def test_complex_elementwise_functions(bb, rng):
    """abs / sqrt / exp / log / angle, Block::operator* and /, max_abs, scale_axis with complex factors, against numpy."""
    z = rng.standard_normal((37, 21)) + 1j * rng.standard_normal((37, 21))
    w = rng.standard_normal((37, 21)) + 1j * rng.standard_normal((37, 21))
    z[0, :6] = [0.0, -4.0, 4.0, 3j, -3j, -1e-300 + 0j]     # branch cuts and signed zeros of sqrt / log / angle
    Z, W = bb.as_block(z), bb.as_block(w)
    tol = dict(rtol=1e-13, atol=1e-14)
    got = bb.abs(Z)
    assert got.dtype == np.dtype('float64')
    np.testing.assert_allclose(bb.to_numpy(got), np.abs(z), **tol)
    np.testing.assert_allclose(bb.to_numpy(bb.angle(Z)), np.angle(z), **tol)
    np.testing.assert_allclose(bb.to_numpy(bb.sqrt(Z)), np.sqrt(z), **tol)
    np.testing.assert_allclose(bb.to_numpy(bb.exp(Z)), np.exp(z), **tol)
    nz = z.copy()
    nz[0, 0] = 1.0
    np.testing.assert_allclose(bb.to_numpy(bb.log(bb.as_block(nz))), np.log(nz), **tol)
    np.testing.assert_allclose(bb.to_numpy(Z * W), z * w, **tol)
    np.testing.assert_allclose(bb.to_numpy(Z / W), z / w, **tol)
    np.testing.assert_allclose(bb.to_numpy(bb.multiply_blocks(Z, bb.as_block(w.real))), z * w.real, **tol)   # mixed dtypes promote
    assert abs(bb.max_abs(Z) - np.abs(z).max()) <= 1e-14 * np.abs(z).max()
    # non-contiguous operands
    np.testing.assert_allclose(bb.to_numpy(bb.permute_axes(Z, [1, 0]) * bb.permute_axes(W, [1, 0])), (z * w).T, **tol)
    f = rng.standard_normal(21) + 1j * rng.standard_normal(21)
    np.testing.assert_allclose(bb.to_numpy(bb.scale_axis(Z, bb.as_block(f), 1)), z * f[None, :], **tol)
    np.testing.assert_allclose(bb.to_numpy(bb.scale_axis(bb.as_block(z.real), bb.as_block(f), 1)), z.real * f[None, :], **tol)
    assert bb.allclose(Z, bb.as_block(z * (1 + 1e-12))) and not bb.allclose(Z, W)
