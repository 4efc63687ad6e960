"""Error behaviour at the C-ABI: invalid arguments are refused with CYB_ERR_INVALID (-> ValueError,
as the reference raises std::invalid_argument -> ValueError, numpy.cpp:1296) BEFORE anything is
launched -- a bad descriptor must never reach a kernel."""

import numpy as np
import pytest

from cyten_amd import _lib

pytestmark = pytest.mark.gpu


def test_gemm_descriptor_validation(bb, rng):
    lib, ctx = bb.lib, bb.ctx.handle
    a = bb.as_block(rng.standard_normal((8, 6)))
    b = bb.as_block(rng.standard_normal((6, 5)))
    c = bb.empty_block((8, 5))
    probs = (_lib.GemmProb * 1)()
    segs = (_lib.GemmSeg * 1)()

    def fill():
        segs[0].A, segs[0].B, segs[0].K = a.ptr, b.ptr, 6
        segs[0].a_rs, segs[0].a_cs, segs[0].b_rs, segs[0].b_cs = 6, 1, 5, 1
        probs[0].C, probs[0].M, probs[0].N, probs[0].ldc = c.ptr, 8, 5, 5
        probs[0].seg_begin, probs[0].seg_end, probs[0].alpha, probs[0].beta = 0, 1, 1.0, 0.0

    fill()
    _lib.check(lib.cyb_gemm_grouped_f64(ctx, probs, 1, segs, 1))
    np.testing.assert_allclose(bb.to_numpy(c), bb.to_numpy(a) @ bb.to_numpy(b), atol=1e-13)
    for mutate, frag in [
        (lambda: setattr(segs[0], 'a_cs', 3) or setattr(segs[0], 'a_rs', 7), 'no unit stride'),
        (lambda: setattr(probs[0], 'ldc', 4), 'ldc'),
        (lambda: setattr(probs[0], 'M', -1), 'bad shape'),
        (lambda: setattr(probs[0], 'seg_end', 5), 'segment range'),
        (lambda: setattr(segs[0], 'K', -2), 'bad K'),
        (lambda: setattr(probs[0], 'C', None), 'C is NULL'),
        (lambda: setattr(segs[0], 'A', None), 'NULL operand'),
    ]:
        fill()
        mutate()
        st = lib.cyb_gemm_grouped_f64(ctx, probs, 1, segs, 1)
        assert st == _lib.CYB_ERR_INVALID
        assert frag in lib.cyb_last_error().decode()
        with pytest.raises(ValueError):
            _lib.check(st)
    # beta != 0 accumulates into C
    fill()
    probs[0].alpha, probs[0].beta = 2.0, 0.5
    before = bb.to_numpy(c)
    _lib.check(lib.cyb_gemm_grouped_f64(ctx, probs, 1, segs, 1))
    np.testing.assert_allclose(bb.to_numpy(c), 2.0 * (bb.to_numpy(a) @ bb.to_numpy(b)) + 0.5 * before, atol=1e-12)


def test_decomposition_and_copy_validation(bb, rng):
    lib, ctx = bb.lib, bb.ctx.handle
    a = bb.as_block(rng.standard_normal((6, 4)))
    U, S, Vh = bb.empty_block((6, 4)), bb.empty_block((4,)), bb.empty_block((4, 4))
    d = (_lib.SvdDesc * 1)()
    d[0].A, d[0].lda, d[0].m, d[0].n = a.ptr, 3, 6, 4            # lda < n
    d[0].U, d[0].ldu, d[0].S, d[0].Vh, d[0].ldvh = U.ptr, 4, S.ptr, Vh.ptr, 4
    assert lib.cyb_svd_batched_f64(ctx, d, 1, None) == _lib.CYB_ERR_INVALID
    d[0].lda, d[0].U = 4, None
    assert lib.cyb_svd_batched_f64(ctx, d, 1, None) == _lib.CYB_ERR_INVALID
    q = (_lib.QrDesc * 1)()
    q[0].A, q[0].lda, q[0].m, q[0].n, q[0].Q, q[0].ldq, q[0].R, q[0].ldr, q[0].full = a.ptr, 4, 6, 4, U.ptr, 2, Vh.ptr, 4, 0
    assert lib.cyb_qr_batched_f64(ctx, q, 1) == _lib.CYB_ERR_INVALID   # ldq < k
    e = (_lib.EighDesc * 1)()
    e[0].A, e[0].lda, e[0].n, e[0].W, e[0].V, e[0].ldv = a.ptr, 2, 4, S.ptr, Vh.ptr, 4
    assert lib.cyb_eigh_batched_f64(ctx, e, 1, None) == _lib.CYB_ERR_INVALID
    cd = (_lib.CopyDesc * 1)()
    cd[0].dst, cd[0].src, cd[0].ndim = U.ptr, a.ptr, 9
    assert lib.cyb_copy_strided_batched(ctx, cd, 1, 8) == _lib.CYB_ERR_INVALID
    cd[0].ndim = 1
    assert lib.cyb_copy_strided_batched(ctx, cd, 1, 3) == _lib.CYB_ERR_INVALID
    assert lib.cyb_svd_batched_f64(None, d, 1, None) == _lib.CYB_ERR_INVALID
    # empty lists are fine
    assert lib.cyb_svd_batched_f64(ctx, None, 0, None) == 0
    assert lib.cyb_gemm_grouped_f64(ctx, None, 0, None, 0) == 0


def test_new_entry_points_validate_arguments(bb, rng):
    """Argument checks of the entries added for truncation, strided linear combinations, extrema, comparisons and the
    complex small-block decompositions: invalid input is an error code (ValueError through the mirror), never a launch."""
    import ctypes as C
    lib, ctx = bb.lib, bb.ctx.handle
    s = bb.as_block(np.sort(rng.random(10))[::-1].copy())
    vd = (_lib.VecDesc * 1)()
    vd[0].x, vd[0].n = s.ptr, 10
    opts = _lib.TruncOpts(5, 0, 0.0, 0.0, 0.0, 0, 1)                  # chi_min < 1
    idx, mask, res = bb.ctx.empty(10, 'int64'), bb.ctx.empty(10, 'uint8'), bb.ctx.empty(3)
    args = (C.c_void_p(idx.data_ptr()), C.c_void_p(mask.data_ptr()), C.c_void_p(res.data_ptr()))
    assert lib.cyb_truncate_select_f64(ctx, vd, 1, C.byref(opts), *args) == _lib.CYB_ERR_INVALID
    opts.chi_min = 1
    assert lib.cyb_truncate_select_f64(ctx, vd, 1, C.byref(opts), None, args[1], args[2]) == _lib.CYB_ERR_INVALID
    assert lib.cyb_truncate_select_f64(ctx, vd, 0, C.byref(opts), *args) == _lib.CYB_ERR_INVALID       # no values
    vd[0].n = 70000                                                                                    # beyond the chunked sort
    assert lib.cyb_truncate_select_f64(ctx, vd, 1, C.byref(opts), *args) == _lib.CYB_ERR_UNSUPPORTED
    ld = (_lib.LincombDesc * 1)()
    lt = (_lib.LincombTerm * 1)()
    ld[0].dst, ld[0].ndim, ld[0].term_begin, ld[0].term_end = s.ptr, 1, 0, 2                           # term range beyond the list
    ld[0].shape[0], ld[0].dst_strides[0] = 10, 1
    lt[0].src, lt[0].coeff = s.ptr, 1.0
    lt[0].src_strides[0] = 1
    assert lib.cyb_lincomb_strided_batched_f64(ctx, ld, 1, lt, 1) == _lib.CYB_ERR_INVALID
    ld[0].term_end, ld[0].ndim = 1, 9
    assert lib.cyb_lincomb_strided_batched_f64(ctx, ld, 1, lt, 1) == _lib.CYB_ERR_INVALID
    out = bb.ctx.empty(2)
    assert lib.cyb_extremum_f64(ctx, C.c_void_p(s.ptr), 0, 0, C.c_void_p(out.data_ptr())) == _lib.CYB_ERR_INVALID
    assert lib.cyb_extremum_f64(ctx, C.c_void_p(s.ptr), 10, 7, C.c_void_p(out.data_ptr())) == _lib.CYB_ERR_INVALID
    assert lib.cyb_compare_f64(ctx, C.c_void_p(s.ptr), None, 0.0, C.c_void_p(mask.data_ptr()), 10, 9) == _lib.CYB_ERR_INVALID
    z = bb.as_block(rng.standard_normal((6, 4)) + 1j * rng.standard_normal((6, 4)))
    U, S, Vh = bb._new((6, 4), True), bb._new((4,)), bb._new((4, 4), True)
    d = (_lib.SvdDesc * 1)()
    d[0].A, d[0].lda, d[0].m, d[0].n = z.ptr, 3, 6, 4                                                  # lda < n
    d[0].U, d[0].ldu, d[0].S, d[0].Vh, d[0].ldvh = U.ptr, 4, S.ptr, Vh.ptr, 4
    assert lib.cyb_svd_batched_c128(ctx, d, 1, None) == _lib.CYB_ERR_INVALID
    d[0].lda, d[0].m, d[0].n, d[0].ldu = 300, 300, 300, 4                                              # ldu < k (large-block path)
    assert lib.cyb_svd_batched_c128(ctx, d, 1, None) == _lib.CYB_ERR_INVALID
    q = (_lib.QrDesc * 1)()
    q[0].A, q[0].lda, q[0].m, q[0].n, q[0].Q, q[0].ldq, q[0].R, q[0].ldr, q[0].full = z.ptr, 4, 6, 4, U.ptr, 2, Vh.ptr, 4, 0
    assert lib.cyb_qr_batched_c128(ctx, q, 1) == _lib.CYB_ERR_INVALID                                 # ldq < k
    assert lib.cyb_svd_batched_c128(ctx, None, 0, None) == 0 and lib.cyb_qr_batched_c128(ctx, None, 0) == 0
    # large-block (device-memory) paths validate the same way before any launch
    big = bb.as_block(rng.standard_normal((200, 150)) + 1j * rng.standard_normal((200, 150)))
    Qb, Rb = bb._new((200, 150), True), bb._new((150, 150), True)
    q[0].A, q[0].lda, q[0].m, q[0].n, q[0].Q, q[0].ldq, q[0].R, q[0].ldr, q[0].full = big.ptr, 150, 200, 150, Qb.ptr, 100, Rb.ptr, 150, 0
    assert lib.cyb_qr_batched_c128(ctx, q, 1) == _lib.CYB_ERR_INVALID                                 # ldq < k
    q[0].ldq, q[0].ldr = 150, 100
    assert lib.cyb_qr_batched_c128(ctx, q, 1) == _lib.CYB_ERR_INVALID                                 # ldr < n
    q[0].ldr, q[0].Q = 150, None
    assert lib.cyb_qr_batched_c128(ctx, q, 1) == _lib.CYB_ERR_INVALID                                 # NULL output
    e = (_lib.EighDesc * 1)()
    hb = bb.as_block(np.eye(150, dtype=complex))
    Wb, Vb = bb._new((150,)), bb._new((150, 150), True)
    e[0].A, e[0].lda, e[0].n, e[0].W, e[0].V, e[0].ldv = hb.ptr, 100, 150, Wb.ptr, Vb.ptr, 150
    assert lib.cyb_eigh_batched_c128(ctx, e, 1, None) == _lib.CYB_ERR_INVALID                         # lda < n
    e[0].lda, e[0].V = 150, None
    assert lib.cyb_eigh_batched_c128(ctx, e, 1, None) == _lib.CYB_ERR_INVALID                         # eigenvectors are required
    e[0].V = Vb.ptr
    assert lib.cyb_eigh_batched_c128(ctx, e, 1, None) == 0
    np.testing.assert_allclose(bb.to_numpy(Wb), np.ones(150), rtol=0, atol=1e-12)


def test_svd_ex_flags_are_validated(bb, rng):
    """cyb_svd_batched_ex_f64: unknown flag bits are refused before anything runs; flags = 0 with rank = NULL is the plain call."""
    import ctypes as C
    a = bb.as_block(rng.standard_normal((60, 50)))
    u, s, vh = bb.empty_block((60, 50)), bb.empty_block((50,)), bb.empty_block((50, 50))
    d = (_lib.SvdDesc * 1)()
    d[0].A, d[0].lda, d[0].m, d[0].n = a.ptr, 50, 60, 50
    d[0].U, d[0].ldu, d[0].S, d[0].Vh, d[0].ldvh = u.ptr, 50, s.ptr, vh.ptr, 50
    with pytest.raises((ValueError, _lib.CybError)) as exc:
        _lib.check(bb.lib.cyb_svd_batched_ex_f64(bb.ctx.handle, d, 1, None, 6, None))
    assert 'flag' in str(exc.value)
    rank = (C.c_int32 * 1)()
    _lib.check(bb.lib.cyb_svd_batched_ex_f64(bb.ctx.handle, d, 1, None, _lib.CYB_SVD_SKIP_NULL_VECTORS, rank))
    assert rank[0] == 50                                   # a full-rank block: nothing to skip
    np.testing.assert_allclose(bb.to_numpy(s), np.linalg.svd(bb.to_numpy(a), compute_uv=False), atol=1e-11)
    # CYB_SVD_EMBEDDED_COMPLEX: the block is a 2m x 2n embedding -- odd extents are refused; on a structured block every
    # singular value comes out twice and the reported rank counts both
    d[0].m, d[0].n = 59, 50
    with pytest.raises((ValueError, _lib.CybError)) as exc:
        _lib.check(bb.lib.cyb_svd_batched_ex_f64(bb.ctx.handle, d, 1, None, _lib.CYB_SVD_EMBEDDED_COMPLEX, None))
    assert 'even' in str(exc.value)
    z = rng.standard_normal((30, 25)) + 1j * rng.standard_normal((30, 25))
    M = np.zeros((60, 50))
    M[0::2, 0::2], M[0::2, 1::2], M[1::2, 0::2], M[1::2, 1::2] = z.real, -z.imag, z.imag, z.real
    a = bb.as_block(M)
    d[0].A, d[0].m, d[0].n = a.ptr, 60, 50
    _lib.check(bb.lib.cyb_svd_batched_ex_f64(bb.ctx.handle, d, 1, None, _lib.CYB_SVD_EMBEDDED_COMPLEX, rank))
    assert rank[0] == 50
    sv = bb.to_numpy(s)
    np.testing.assert_allclose(sv[0::2], np.linalg.svd(z, compute_uv=False), atol=1e-11)
    np.testing.assert_allclose(sv[1::2], sv[0::2], atol=1e-12)
    U, Vh = bb.to_numpy(u), bb.to_numpy(vh)
    np.testing.assert_allclose((U * sv) @ Vh, M, atol=1e-11)
    # structure: real column 2a+1 of U is the embedding partner of column 2a (and the same for the rows of Vh)
    np.testing.assert_allclose(U[0::2, 1::2], -U[1::2, 0::2], atol=1e-11)
    np.testing.assert_allclose(U[1::2, 1::2], U[0::2, 0::2], atol=1e-11)


def test_eigh_ex_flags_are_validated(bb, rng):
    """cyb_eigh_batched_ex_f64: unknown flag bits and odd extents of an embedded block are refused; on the embedding of a
    Hermitian block every eigenvalue comes out twice and real column 2a + 1 of V is the embedding partner of column 2a."""
    z = rng.standard_normal((40, 40)) + 1j * rng.standard_normal((40, 40))
    h = z + z.conj().T
    M = np.zeros((80, 80))
    M[0::2, 0::2], M[0::2, 1::2], M[1::2, 0::2], M[1::2, 1::2] = h.real, -h.imag, h.imag, h.real
    a = bb.as_block(M)
    w, v = bb.empty_block((80,)), bb.empty_block((80, 80))
    d = (_lib.EighDesc * 1)()
    d[0].A, d[0].lda, d[0].n, d[0].W, d[0].V, d[0].ldv = a.ptr, 80, 80, w.ptr, v.ptr, 80
    with pytest.raises((ValueError, _lib.CybError)) as exc:
        _lib.check(bb.lib.cyb_eigh_batched_ex_f64(bb.ctx.handle, d, 1, None, 5))
    assert 'flag' in str(exc.value)
    d[0].n = 79
    with pytest.raises((ValueError, _lib.CybError)) as exc:
        _lib.check(bb.lib.cyb_eigh_batched_ex_f64(bb.ctx.handle, d, 1, None, _lib.CYB_EIGH_EMBEDDED_COMPLEX))
    assert 'even' in str(exc.value)
    d[0].n = 80
    _lib.check(bb.lib.cyb_eigh_batched_ex_f64(bb.ctx.handle, d, 1, None, _lib.CYB_EIGH_EMBEDDED_COMPLEX))
    W, V = bb.to_numpy(w), bb.to_numpy(v)
    np.testing.assert_allclose(W[0::2], np.linalg.eigvalsh(h), atol=1e-11 * 40)
    np.testing.assert_allclose(W[1::2], W[0::2], atol=1e-12 * 40)
    np.testing.assert_allclose(M @ V, V * W, atol=1e-10)
    np.testing.assert_allclose(V[0::2, 1::2], -V[1::2, 0::2], atol=1e-11)
    np.testing.assert_allclose(V[1::2, 1::2], V[0::2, 0::2], atol=1e-11)
    _lib.check(bb.lib.cyb_eigh_batched_ex_f64(bb.ctx.handle, d, 1, None, 0))        # flags = 0: the plain call
    np.testing.assert_allclose(bb.to_numpy(w), np.linalg.eigvalsh(M), atol=1e-10)
