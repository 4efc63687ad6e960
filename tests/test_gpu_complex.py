"""complex128 blocks on the tdot path (the reference's second dtype, numpy.cpp dispatches every virtual on
it): storage / data movement / BLAS-1 / grouped GEMM through the real f64-MFMA kernel, against numpy.
Decompositions of complex blocks are not on the device path yet and must say so."""
import os

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


def _csvd_check(a, U, S, Vh, tol=1e-10):
    """The reference's SVD acceptance criteria (test_tensors.py:3405-3500) for complex blocks, plus LAPACK's values."""
    k = min(a.shape)
    assert U.shape == (a.shape[0], k) and S.shape == (k,) and Vh.shape == (k, a.shape[1])
    assert S.dtype == np.float64 and U.dtype == np.complex128 and Vh.dtype == np.complex128
    sc = np.abs(a).max() or 1.0          # compare at unit scale: norms of entries near 1e+-250 leave the double range
    a, S = a / sc, S / sc
    nrm = max(np.linalg.norm(a), 1e-300)
    assert np.all(S >= 0) and np.all(S[:-1] >= S[1:] - tol * nrm)
    assert np.abs((U * S) @ Vh - a).max() <= tol * nrm
    assert np.abs(U.conj().T @ U - np.eye(k)).max() <= tol
    assert np.abs(Vh @ Vh.conj().T - np.eye(k)).max() <= tol
    assert np.abs(S - np.linalg.svd(a, compute_uv=False)).max() <= tol * nrm


def test_complex_svd_small_blocks(bb, rng):
    """complex128 SVD through the in-LDS complex Jacobi kernel: tall / wide / odd shapes, rank deficiency with
    completion of the null directions, zero and identity blocks, purely imaginary and real-valued complex blocks, a
    mixed real / complex list (promoted), a batch of 200 random small blocks."""
    shapes = [(1, 1), (2, 3), (3, 2), (6, 6), (17, 9), (9, 17), (40, 40), (64, 64), (63, 31), (31, 63), (96, 48), (48, 96), (128, 20)]
    mats = [crandn(rng, s) for s in shapes]
    low = crandn(rng, (30, 3)) @ crandn(rng, (3, 22))
    mats += [low, low.T.copy(), np.zeros((7, 4), complex), np.eye(9, dtype=complex), 1j * rng.standard_normal((8, 8)),
             rng.standard_normal((10, 6)).astype(complex), np.ones((12, 5)) * (1 + 1j), 1e-80 * crandn(rng, (6, 9)), 1e200 * crandn(rng, (9, 6)),
             1e-250 * crandn(rng, (5, 5))]
    res = bb.matrix_svd_batched([bb.as_block(m) for m in mats])
    for m, (u, s, vh) in zip(mats, res):
        _csvd_check(m, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh))
    real = rng.standard_normal((11, 7))
    (u1, s1, v1), (u2, s2, v2) = bb.matrix_svd_batched([bb.as_block(real), bb.as_block(mats[4])])
    _csvd_check(real.astype(complex), bb.to_numpy(u1), bb.to_numpy(s1), bb.to_numpy(v1))
    many = [crandn(rng, (int(rng.integers(1, 33)), int(rng.integers(1, 33)))) for _ in range(200)]
    for m, (u, s, vh) in zip(many, bb.matrix_svd_batched([bb.as_block(m) for m in many])):
        _csvd_check(m, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh))
    u, s, vh = bb.matrix_svd(bb.permute_axes(bb.as_block(mats[4]), [1, 0]))       # a transposed view
    _csvd_check(mats[4].T, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh))


def test_complex_eigh_small_blocks(bb, rng):
    """Hermitian eigh of complex128 blocks (np.linalg.eigh semantics: ascending eigenvalues, unitary eigenvectors),
    including degenerate and indefinite spectra, a negative definite block and a real symmetric block typed complex."""
    mats = []
    for n in (1, 2, 5, 16, 33, 64):
        z = crandn(rng, (n, n))
        mats.append(z + z.conj().T)
    q, _ = np.linalg.qr(crandn(rng, (12, 12)))
    mats.append((q * np.array([-3.0] * 4 + [0.0] * 4 + [2.0] * 4)) @ q.conj().T)     # three four-fold eigenvalues
    mats.append(-(mats[3] @ mats[3].conj().T) - np.eye(16))                           # negative definite
    mats.append(np.zeros((6, 6), complex))
    r = rng.standard_normal((9, 9))
    mats.append((r + r.T).astype(complex))
    for h, (w, v) in zip(mats, bb.eigh_batched([bb.as_block(h) for h in mats])):
        w, v = bb.to_numpy(w), bb.to_numpy(v)
        nrm = max(np.abs(h).max(), 1e-300) * h.shape[0]
        assert w.dtype == np.float64 and v.dtype == np.complex128
        assert np.all(np.diff(w) >= -1e-10 * nrm)
        assert np.abs(w - np.linalg.eigvalsh(h)).max() <= 1e-10 * nrm
        assert np.abs(h @ v - v * w).max() <= 1e-10 * nrm
        assert np.abs(v.conj().T @ v - np.eye(h.shape[0])).max() <= 1e-10
    w = bb.eigvalsh(bb.as_block(mats[2]))
    assert np.abs(bb.to_numpy(w) - np.linalg.eigvalsh(mats[2])).max() <= 1e-10 * np.abs(mats[2]).max() * 5
    w, v = bb.eigh(bb.as_block(mats[2]), sort='>')
    assert np.all(np.diff(bb.to_numpy(w)) <= 1e-12)


@pytest.mark.parametrize('full', [False, True])
def test_complex_qr_lq_small_blocks(bb, rng, full):
    """scipy.linalg.qr(mode='economic' | 'full') semantics on complex128 blocks: A = Q R, Q^H Q = 1, R upper triangular
    (test_qr_lq of the reference, test_tensors.py:3166-), for tall / wide / square blocks, dependent and zero columns."""
    shapes = [(1, 1), (5, 3), (3, 5), (16, 16), (40, 17), (17, 40), (64, 64), (96, 30), (30, 96), (128, 20)]
    mats = [crandn(rng, s) for s in shapes]
    dep = crandn(rng, (20, 8))
    dep[:, 3] = dep[:, 1] * (0.5 - 2j)
    dep[:, 6] = 0.0
    mats += [dep, crandn(rng, (12, 2)) @ crandn(rng, (2, 9)), np.zeros((6, 4), complex), rng.standard_normal((7, 7)).astype(complex)]
    mats = [m for m in mats if not full or m.shape[0] <= 96]
    for a, (q, r) in zip(mats, bb.matrix_qr_batched([bb.as_block(m) for m in mats], full)):
        q, r = bb.to_numpy(q), bb.to_numpy(r)
        m, n = a.shape
        kq = m if full else min(m, n)
        assert q.shape == (m, kq) and r.shape == (kq, n) and q.dtype == np.complex128
        nrm = max(np.abs(a).max(), 1e-300) * max(m, n)
        assert np.abs(q @ r - a).max() <= 1e-10 * nrm
        assert np.abs(q.conj().T @ q - np.eye(kq)).max() <= 1e-10
        assert np.abs(np.tril(r, -1)).max() == 0.0
    a = mats[4]
    l, q = bb.matrix_lq(bb.as_block(a), full)
    l, q = bb.to_numpy(l), bb.to_numpy(q)
    assert np.abs(l @ q - a).max() <= 1e-10 * np.abs(a).max() * max(a.shape)
    assert np.abs(q @ q.conj().T - np.eye(q.shape[0])).max() <= 1e-10 and np.abs(np.triu(l, 1)).max() == 0.0


@pytest.mark.parametrize('full', [False, True])
def test_complex_qr_lq_large_blocks(bb, rng, full):
    """Blocks beyond the in-LDS limit: blocked CGS2 in device memory (csrc/csvd_large.hip) -- tall / wide / square, extents
    that are not multiples of the panel width, dependent and zero columns (completion), extreme scales, mixed with small
    blocks in one list."""
    shapes = [(200, 200), (301, 77), (77, 301), (129, 130), (257, 31), (150, 16), (140, 400)]
    mats = [crandn(rng, s) for s in shapes]
    dep = crandn(rng, (160, 60))
    dep[:, 17] = dep[:, 3] * (0.5 - 2j) + dep[:, 9]
    dep[:, 40] = 0.0
    wide_dep = crandn(rng, (130, 20)) @ crandn(rng, (20, 300))           # rank 20 < m: most leading columns are dependent
    mats += [dep, crandn(rng, (170, 30)) @ crandn(rng, (30, 140)), wide_dep, np.zeros((140, 9), complex), 1e-150 * crandn(rng, (150, 40)),
             1e150 * crandn(rng, (40, 150)), crandn(rng, (12, 7))]
    for a, (q, r) in zip(mats, bb.matrix_qr_batched([bb.as_block(m) for m in mats], full)):
        q, r = bb.to_numpy(q), bb.to_numpy(r)
        m, n = a.shape
        kq = m if full else min(m, n)
        assert q.shape == (m, kq) and r.shape == (kq, n) and q.dtype == np.complex128
        sc = np.abs(a).max() or 1.0
        assert np.abs(q @ (r / sc) - a / sc).max() <= 1e-10 * max(m, n)
        assert np.abs(q.conj().T @ q - np.eye(kq)).max() <= 1e-10
        assert np.abs(np.tril(r, -1)).max() == 0.0
    a = mats[1]
    l, q = bb.matrix_lq(bb.as_block(a), full)
    l, q = bb.to_numpy(l), bb.to_numpy(q)
    assert np.abs(l @ q - a).max() <= 1e-10 * np.abs(a).max() * max(a.shape)
    assert np.abs(q @ q.conj().T - np.eye(q.shape[0])).max() <= 1e-10 and np.abs(np.triu(l, 1)).max() == 0.0


def test_complex_svd_large_blocks(bb, rng):
    """complex128 blocks beyond the in-LDS limit go through the device-memory Jacobi (csrc/csvd_large.hip): square, tall,
    wide and odd extents, a rank-deficient product (block completion of the null directions), a block with ONE null
    direction, degenerate singular values, a zero block, extreme scales, and a list that mixes small and large blocks."""
    shapes = [(65, 65), (200, 200), (301, 77), (77, 301), (129, 130), (257, 255)]
    mats = [crandn(rng, s) for s in shapes]
    low = crandn(rng, (180, 40)) @ crandn(rng, (40, 150))                # rank 40 of 150
    one = crandn(rng, (90, 89)) @ crandn(rng, (89, 90))                  # rank 89 of 90
    q1, _ = np.linalg.qr(crandn(rng, (140, 100)))
    q2, _ = np.linalg.qr(crandn(rng, (100, 100)))
    deg = (q1 * np.repeat([3.0, 2.0, 2.0, 1.0, 0.5], 20)) @ q2.conj().T   # twenty-fold singular values
    mats += [low, low.conj().T.copy(), one, deg, np.zeros((70, 130), complex), 1e-150 * crandn(rng, (100, 66)), 1e150 * crandn(rng, (66, 100)),
             crandn(rng, (12, 7))]
    res = bb.matrix_svd_batched([bb.as_block(m) for m in mats])
    for m, (u, s, vh) in zip(mats, res):
        _csvd_check(m, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh))
    u, s, vh = bb.matrix_svd(bb.permute_axes(bb.as_block(mats[2]), [1, 0]))       # a transposed view
    _csvd_check(mats[2].T, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh))


def _sparse_product(rng, m, n, cplx):
    """A block as a composed block-sparse tensor has them: a product of factors with zero sub-blocks, zero columns -- the R of its
    QR is a staircase whose numerically zero rows sit in the MIDDLE (scripts/svd_fuzz.py)."""
    g = (lambda sh: crandn(rng, sh)) if cplx else rng.standard_normal
    r = max(2, min(m, n) // 2)
    b1, b2 = g((m, r)), g((r, n))
    b1[rng.random((m, 1)) < 0.5 * np.ones((1, r)) * (np.arange(r) % 2)] = 0.0
    b2[:, rng.random(n) < 0.4] = 0.0
    b2[np.arange(r) % 3 == 0, : n // 2] = 0.0
    return b1 @ b2


def test_complex_svd_of_blocks_with_dependent_columns_in_the_middle(bb, rng):
    """The embedded route needs R = M(R_c) structured, i.e. independent leading columns; zero columns / products of block-sparse
    factors break that (singular values off by 1e-3 or no convergence before the reconstruction check was added: the blocks
    it flags go to the complex kernels).  Found by `scripts/svd_fuzz.py`, seeds 52 / 53."""
    mats = [_sparse_product(rng, 455, 440, True), _sparse_product(rng, 139, 2017, True), _sparse_product(rng, 600, 130, True)]
    z = crandn(rng, (300, 260))
    z[:, rng.random(260) < 0.4] = 0.0
    mats.append(z)
    z = crandn(rng, (260, 300))
    z[rng.random(260) < 0.3] = 0.0
    mats.append(z)
    mats.append(crandn(rng, (200, 150)))
    res, info = bb.matrix_svd_batched([bb.as_block(m) for m in mats], return_info=True)
    for m, (u, s, vh) in zip(mats, res):
        _csvd_check(m, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh))
    res2, ranks = bb.matrix_svd_batched([bb.as_block(m) for m in mats], null_vectors=False, return_rank=True)
    for m, (u, s, vh), rk in zip(mats, res2, ranks):
        u, s, vh = bb.to_numpy(u)[:, :rk], bb.to_numpy(s), bb.to_numpy(vh)[:rk]
        nrm = np.linalg.norm(m)
        assert np.abs(s - np.linalg.svd(m, compute_uv=False)).max() <= 1e-10 * nrm and np.abs((u * s[:rk]) @ vh - m).max() <= 1e-10 * nrm
        assert np.abs(u.conj().T @ u - np.eye(rk)).max() <= 1e-10 and np.abs(vh @ vh.conj().T - np.eye(rk)).max() <= 1e-10


def _copied_column_blocks(rng, shapes):
    mats = []
    for m, n in shapes:
        r = max(min(m, n) // 3, 1)                           # (the blocks that failed: rank k/3 AND copied columns from column ~r/2 on)
        a = crandn(rng, (m, r)) @ crandn(rng, (r, n))
        src, dst = rng.integers(0, n, max(n // 5, 1)), rng.integers(0, n, max(n // 5, 1))
        a[:, dst] = a[:, src] * rng.integers(1, 4, len(src))
        mats.append(a)
        k = min(m, n)
        p = np.zeros((m, n), complex)
        p[rng.permutation(m)[:k], rng.permutation(n)[:k]] = rng.choice([1.0, 2.0, 0.5, 3.0], k) * crandn(rng, (k,))
        p = p * (rng.random(n) < 0.9)
        for r in rng.integers(0, m, 3):
            p[r] = crandn(rng, (n,))
        mats.append(p)
    return mats


def _cqr_check(a, q, r, full):
    m, n = a.shape
    kq = m if full else min(m, n)
    assert q.shape == (m, kq) and r.shape == (kq, n)
    nrm = max(np.linalg.norm(a), 1e-300)
    assert np.abs(q @ r - a).max() <= 1e-10 * nrm
    assert np.abs(q.conj().T @ q - np.eye(kq)).max() <= 1e-10
    assert np.abs(np.tril(r, -1)).max() == 0.0
    d = np.diagonal(r)
    assert np.abs(d.imag).max(initial=0.0) == 0.0 and d.real.min(initial=0.0) >= 0.0   # both routes: diag(R) real, >= 0


@pytest.mark.parametrize('full', [False, True])
def test_complex_qr_with_copied_columns(bb, rng, full):
    """Exact copies of columns / scaled partial permutations in low-rank complex blocks (`scripts/svd_fuzz.py`, seeds 93 - 95):
    the embedded route gives such blocks up (a dependent column in the middle) and the Gram-Schmidt kernels that used to stand
    behind it returned a non-unitary Q for some of them (round 2's open defect).  Every block is now factored -- the embedded
    route where its checks pass, complex Householder QR (csrc/cqr_house.hip) otherwise, block by block: A = Q R, Q^H Q = 1,
    R upper triangular with a non-negative real diagonal, as scipy.linalg.qr factors them (numpy.cpp:1236-1245).  Large blocks,
    blocks inside the limits of the one-workgroup kernel (min <= 48, max <= 128) and the soak's own 182 x 661 block."""
    mats = _copied_column_blocks(rng, [(678, 551), (413, 513), (182, 661), (150, 150), (40, 128), (128, 33), (48, 48), (17, 90), (64, 64),
                                       (5, 3), (96, 100)])
    mats.append(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'cqr_fuzz_seed95_list31_182x661.npz'))['a'])
    res = bb.matrix_qr_batched([bb.as_block(a) for a in mats], full)           # one list: per-block routing
    for a, (q, r) in zip(mats, res):
        _cqr_check(a, bb.to_numpy(q), bb.to_numpy(r), full)
    for a in mats[:6] + mats[-1:]:                                              # and one block per call
        (q, r), = bb.matrix_qr_batched([bb.as_block(a)], full)
        _cqr_check(a, bb.to_numpy(q), bb.to_numpy(r), full)
    # LQ of the transposes (block_backend.cpp:1033-1040)
    for a, (l, q) in zip(mats[:8], bb.matrix_lq_batched([bb.as_block(a.T.copy()) for a in mats[:8]], full)):
        l, q = bb.to_numpy(l), bb.to_numpy(q)
        at = a.T
        assert np.abs(l @ q - at).max() <= 1e-10 * np.linalg.norm(at) and np.abs(q @ q.conj().T - np.eye(q.shape[0])).max() <= 1e-10
        assert np.abs(np.triu(l, 1)).max() == 0.0


@pytest.mark.parametrize('full', [False, True])
def test_complex_householder_qr_kernels_directly(bb, rng, full):
    """`cyb_qr_batched_c128` itself (no embedded route in front): the one-workgroup kernel and the launch-per-step path on full-rank,
    rank-deficient, zero, single-column, wide, tall and extreme-scale blocks, against the acceptance criteria of the reference's
    test_qr_lq (tests/python_tests/test_tensors.py:3166) and R against scipy.linalg.qr up to the row signs."""
    import scipy.linalg
    mats = [crandn(rng, s) for s in [(1, 1), (1, 7), (7, 1), (33, 20), (20, 33), (64, 64), (96, 96), (97, 130), (130, 97), (300, 40), (40, 300),
                                     (257, 255)]]
    mats += [np.zeros((30, 20), complex), np.ones((50, 60), complex), crandn(rng, (80, 10)) @ crandn(rng, (10, 70)),
             1e150 * crandn(rng, (60, 40)), 1e-150 * crandn(rng, (40, 60)), np.eye(70, 50).astype(complex) * 1j]
    mats += _copied_column_blocks(rng, [(150, 150), (40, 128)])
    srcs = bb.contiguous_many([bb.as_block(a) for a in mats])
    res = bb.matrix_qr_batched_direct(srcs, full, True)
    for a, (q, r) in zip(mats, res):
        q, r = bb.to_numpy(q), bb.to_numpy(r)
        _cqr_check(a, q, r, full)
        if np.linalg.matrix_rank(a) == min(a.shape):      # unique up to the signs of R's rows
            rs = scipy.linalg.qr(a, mode='full' if full else 'economic')[1]
            k = min(a.shape)
            ph = np.diagonal(rs)[:k] / np.abs(np.diagonal(rs)[:k])
            assert np.abs(r[:k] - rs[:k] / ph[:, None]).max() <= 1e-10 * np.linalg.norm(a)


def test_complex_svd_embedded_route(bb, rng):
    """Blocks with min(m, n) >= 96 are decomposed by the float64 block engine on their interleaved embeddings
    (`cyb_svd_batched_ex_f64` with CYB_SVD_EMBEDDED_COMPLEX: structured pivot solves, pair-wise deflation, sign-consistent
    reflectors).  The route itself (not the complex Jacobi kernels behind it): every singular value once, degenerate
    singular subspaces (a unitary block: ONE 150-fold value), null spaces on either side, entries near 1e+-150 (the
    range-scaled copy keeps the flag), 48 ... 700 rows in one list -- and the values of the complex kernels beside them."""
    q, _ = np.linalg.qr(crandn(rng, (150, 150)))
    q1, _ = np.linalg.qr(crandn(rng, (140, 100)))
    q2, _ = np.linalg.qr(crandn(rng, (100, 100)))
    mats = [q, (q1 * np.repeat([3.0, 2.0, 2.0, 1.0, 0.5], 20)) @ q2.conj().T, crandn(rng, (48, 48)), crandn(rng, (49, 333)),
            crandn(rng, (700, 64)), crandn(rng, (100, 1)) @ crandn(rng, (1, 90)), crandn(rng, (180, 40)) @ crandn(rng, (40, 150)),
            crandn(rng, (150, 40)) @ crandn(rng, (40, 180)), crandn(rng, (90, 89)) @ crandn(rng, (89, 90)), np.zeros((60, 70), complex),
            1e-150 * crandn(rng, (100, 66)), 1e150 * crandn(rng, (66, 100)), crandn(rng, (256, 256)) * np.logspace(0, -12, 256)]
    srcs = bb.contiguous_many([bb.as_block(m) for m in mats])
    got = bb._complex_svd_embedded(srcs, return_info=True)
    assert got is not None
    res, info, ranks = got
    assert ranks[:2] == [150, 100] and ranks[5:8] == [1, 40, 40] and ranks[8] in (89, 90) and ranks[9] == 0, ranks
    direct = bb.matrix_svd_batched_complex_direct(srcs)
    for m, (u, s, vh), (_, sd, _) in zip(mats, res, direct):
        _csvd_check(m, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh))
        assert np.abs(bb.to_numpy(s) - bb.to_numpy(sd)).max() <= 1e-10 * max(np.linalg.norm(m), 1e-300)
    # (the twenty-fold values converge linearly -- rotations inside a degenerate cluster are 45 degrees however small the
    #  coupling -- exactly as their real embedding does on the real engine: 26 sweeps either way)
    assert max(info[:1] + info[2:]) <= 14 and info[1] <= 32, info
    # the truncating caller's form (null vectors skipped, numerical ranks reported): the leading rank triplets are those of
    # the full call, orthonormal in the complex sense, and reconstruct the block
    low = [mats[6], mats[7], mats[8], mats[4]]
    res2, ranks2 = bb.matrix_svd_batched([bb.as_block(m) for m in low], null_vectors=False, return_rank=True)
    assert ranks2[:2] == [40, 40] and ranks2[2] in (89, 90) and ranks2[3] == 64, ranks2   # (sigma_90 of the third sits AT the threshold)
    for m, (u, s, vh), r in zip(low, res2, ranks2):
        u, s, vh = bb.to_numpy(u)[:, :r], bb.to_numpy(s), bb.to_numpy(vh)[:r]
        nrm = np.linalg.norm(m)
        assert np.abs((u * s[:r]) @ vh - m).max() <= 1e-10 * nrm and np.abs(s - np.linalg.svd(m, compute_uv=False)).max() <= 1e-10 * nrm
        assert np.abs(u.conj().T @ u - np.eye(r)).max() <= 1e-10 and np.abs(vh @ vh.conj().T - np.eye(r)).max() <= 1e-10
    # the public entry takes this route for the large blocks of a mixed list and the in-LDS kernel for the small ones
    mixed = [mats[2], crandn(rng, (12, 7)), mats[6], crandn(rng, (30, 30))]
    for m, (u, s, vh) in zip(mixed, bb.matrix_svd_batched([bb.as_block(m) for m in mixed])):
        _csvd_check(m, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh))


@pytest.mark.parametrize('full', [False, True])
def test_complex_qr_of_low_rank_plus_noise_blocks(bb, rng, full):
    """Found by `scripts/svd_fuzz.py`: a low-rank block plus noise at 1e-5 ... 1e-12 of its norm (what a two-site theta with
    a little numerical dirt is).  The Gram-Schmidt kernels of round 1 returned a Q with |Q^H Q - 1| ~ 1 for it, and the embedded
    route loses the structure of its trailing reflectors (defect eps |A| / sigma): blocks beyond the in-LDS kernel now always
    take the embedded route, whose factor is checked and factored a second time where needed (`_complex_qr_embedded`)."""
    mats = []
    for (m, n), noise in [((300, 200), 1e-5), ((300, 200), 1e-9), ((200, 300), 1e-9), ((588, 532), 1e-12), ((161, 159), 1e-7), ((70, 400), 1e-8),
                          ((1989, 106), 1e-9)]:
        r = min(m, n) // 3
        mats.append(crandn(rng, (m, r)) @ crandn(rng, (r, n)) + noise * crandn(rng, (m, n)))
    mats.append(crandn(rng, (255, 431)) * np.logspace(0, -12, 431))           # graded columns
    for a, (q, r) in zip(mats, bb.matrix_qr_batched([bb.as_block(a) for a in mats], full)):
        q, r = bb.to_numpy(q), bb.to_numpy(r)
        m, n = a.shape
        kq = m if full else min(m, n)
        assert q.shape == (m, kq) and r.shape == (kq, n)
        assert np.abs(q @ r - a).max() <= 1e-10 * np.linalg.norm(a)
        assert np.abs(q.conj().T @ q - np.eye(kq)).max() <= 1e-10
        assert np.abs(np.tril(r, -1)).max() <= 1e-10 * np.linalg.norm(a)


def test_complex_eigh_large_blocks(bb, rng):
    mats = []
    for n in (65, 150, 257):
        z = crandn(rng, (n, n))
        mats.append(z + z.conj().T)
    q, _ = np.linalg.qr(crandn(rng, (120, 120)))
    mats.append((q * np.repeat([-3.0, 0.0, 2.0], 40)) @ q.conj().T)                   # three forty-fold eigenvalues
    mats.append(-(mats[0] @ mats[0].conj().T) - np.eye(65))                           # negative definite
    mats.append(np.zeros((80, 80), complex))
    mats.append(mats[0][:9, :9].copy())                                               # a small one in the same list
    for h, (w, v) in zip(mats, bb.eigh_batched([bb.as_block(h) for h in mats])):
        w, v = bb.to_numpy(w), bb.to_numpy(v)
        nrm = max(np.abs(h).max(), 1e-300) * h.shape[0]
        assert w.dtype == np.float64 and v.dtype == np.complex128
        assert np.all(np.diff(w) >= -1e-10 * nrm)
        assert np.abs(w - np.linalg.eigvalsh(h)).max() <= 1e-10 * nrm
        assert np.abs(h @ v - v * w).max() <= 1e-10 * nrm
        assert np.abs(v.conj().T @ v - np.eye(h.shape[0])).max() <= 1e-10


def test_complex_eigh_embedded_route(bb, rng):
    """Hermitian blocks with n >= 96 are diagonalised by the float64 block engine on their interleaved embeddings
    (`cyb_eigh_batched_ex_f64` with CYB_EIGH_EMBEDDED_COMPLEX).  The route itself, from n = 48: every eigenvalue once and
    ascending, unitary eigenvector matrices also inside forty-fold eigenvalues, indefinite / negative definite / zero /
    real-valued / diagonal blocks, entries near 1e+-150, and the values of the complex kernels beside them."""
    mats = []
    for n in (48, 97, 150, 257, 400):
        z = crandn(rng, (n, n))
        mats.append(z + z.conj().T)
    q, _ = np.linalg.qr(crandn(rng, (120, 120)))
    mats.append((q * np.repeat([-3.0, 0.0, 2.0], 40)) @ q.conj().T)
    mats[-1] = 0.5 * (mats[-1] + mats[-1].conj().T)
    mats.append(-(mats[1] @ mats[1].conj().T) - np.eye(97))
    mats.append(np.zeros((80, 80), complex))
    r = rng.standard_normal((100, 100))
    mats.append((r + r.T).astype(complex))
    mats.append(np.diag(rng.standard_normal(64)).astype(complex))
    mats += [1e-150 * mats[2], 1e150 * mats[1]]
    srcs = bb.contiguous_many([bb.as_block(h) for h in mats])
    got = bb._complex_eigh_embedded(srcs, return_info=True)
    assert got is not None
    direct = bb.eigh_batched(srcs, _embed=False)
    for h, (w, v), (wd, _) in zip(mats, got[0], direct):
        w, v = bb.to_numpy(w), bb.to_numpy(v)
        sc = np.abs(h).max() or 1.0
        h, w, wd = h / sc, w / sc, bb.to_numpy(wd) / sc
        nrm = max(np.abs(h).max(), 1e-300) * h.shape[0]
        assert w.dtype == np.float64 and v.dtype == np.complex128 and w.shape == (h.shape[0],) and v.shape == h.shape
        assert np.all(np.diff(w) >= 0)
        assert np.abs(w - np.linalg.eigvalsh(h)).max() <= 1e-10 * nrm and np.abs(w - wd).max() <= 1e-10 * nrm
        assert np.abs(h @ v - v * w).max() <= 1e-10 * nrm
        assert np.abs(v.conj().T @ v - np.eye(h.shape[0])).max() <= 1e-10
    # the public entries: a mixed list (embedded route for the large blocks, in-LDS kernel for the small one), eigvalsh, sort
    mixed = [mats[3], mats[0][:9, :9].copy(), mats[5]]
    for h, (w, v) in zip(mixed, bb.eigh_batched([bb.as_block(h) for h in mixed])):
        w, v = bb.to_numpy(w), bb.to_numpy(v)
        assert np.abs(h @ v - v * w).max() <= 1e-10 * np.abs(h).max() * h.shape[0]
    w = bb.to_numpy(bb.eigvalsh(bb.as_block(mats[3])))
    assert np.abs(w - np.linalg.eigvalsh(mats[3])).max() <= 1e-10 * np.abs(mats[3]).max() * 257
    w, v = bb.eigh(bb.as_block(mats[2]), sort='>')
    assert np.all(np.diff(bb.to_numpy(w)) <= 0) and np.abs(mats[2] @ bb.to_numpy(v) - bb.to_numpy(v) * bb.to_numpy(w)).max() <= 1e-9


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


@pytest.mark.parametrize('chi_full,chi', [(48, 30), (256, 150)])
def test_complex_theta_truncated_svd_end_to_end(bb, rng, chi_full, chi):
    """The whole hot path in complex arithmetic, at a size whose sector blocks fit the in-LDS kernels and at one whose
    rank-deficient blocks go through the device-memory Jacobi with block completion: U(1) theta = A.B with complex
    blocks, combine_legs, batched complex SVD, truncation on the device, mask gather -- against the dense theta: kept
    singular values = the largest ones of the dense matrix, U S Vh = best rank-chi approximation."""
    A, B = wl.config_u1_mps(chi_full)
    for t in (A, B):
        t.blocks = [b + 1j * rng.standard_normal(b.shape) for b in t.blocks]
    a, b = ab.AbelianTensor.from_spec(bb, A), ab.AbelianTensor.from_spec(bb, B)
    theta = ab.compose(bb, a, b, 1)
    mv, U, S, Vh, err, new_norm = ab.truncated_svd(bb, theta, 2, chi_max=chi)
    dense = theta.to_dense(bb)
    mat = dense.reshape(dense.shape[0] * dense.shape[1], -1)
    s_all = np.linalg.svd(mat, compute_uv=False)
    kept = np.sort(np.concatenate([bb.to_numpy(s) for s in S]))[::-1]
    assert len(kept) == chi and np.abs(kept - s_all[:chi]).max() <= TOL * s_all[0]
    assert abs(new_norm - np.sum(s_all[:chi] ** 2)) <= TOL * np.sum(s_all ** 2)
    assert abs(err - np.sum(s_all[chi:] ** 2)) <= TOL * np.sum(s_all ** 2)
    resid2 = 0.0
    for m, u, s, vh in zip(mv.blocks, U, S, Vh):
        m, u, s, vh = bb.to_numpy(m), bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
        assert u.dtype == np.complex128 and s.dtype == np.float64
        if len(s):   # (a sector may keep nothing)
            assert np.abs(u.conj().T @ u - np.eye(len(s))).max() <= TOL and np.abs(vh @ vh.conj().T - np.eye(len(s))).max() <= TOL
        resid2 += np.linalg.norm(m - (u * s) @ vh) ** 2
    assert abs(resid2 - err) <= 1e-9 * np.sum(s_all ** 2)
