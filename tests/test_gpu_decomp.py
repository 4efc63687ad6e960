"""Parity of the batched SVD / QR / eigh kernels (through the C-ABI) with the oracle (LAPACK via
scipy/numpy, the routines the reference's NumpyBlockBackend calls).  As in the reference's own
tests, U/V/Q entries are not compared (sign / rotation freedom); singular values, eigenvalues,
reconstructions and isometry are, to 1e-10 (BASELINE.json tolerance)."""
import os

import numpy as np
import pytest

from helpers import check_svd_invariants
from oracle import block_ops as ops

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _svd_batch(bb, mats, **kw):
    res = bb.matrix_svd_batched([bb.as_block(m) for m in mats], **kw)
    return [(bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(v)) for u, s, v in res]


def test_svd_ragged_batch(bb, rng):
    shapes = [(1, 1), (5, 3), (3, 5), (1, 9), (9, 1), (32, 32), (33, 31), (64, 64), (65, 64), (100, 37), (37, 100),
              (128, 130), (200, 300), (257, 64)]
    mats = [rng.standard_normal(s) for s in shapes]
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])


def test_svd_rank_deficient_zero_and_graded(bb, rng):
    lowrank = rng.standard_normal((80, 3)) @ rng.standard_normal((3, 60))
    zero = np.zeros((20, 12))
    ident = np.eye(40)
    graded = rng.standard_normal((70, 70)) * np.logspace(0, -14, 70)[None, :]
    exp_decay = (np.linalg.qr(rng.standard_normal((90, 90)))[0] * np.exp(-np.arange(90.0))) @ \
        np.linalg.qr(rng.standard_normal((90, 90)))[0]      # DMRG-like spectrum
    rank1 = np.outer(rng.standard_normal(50), rng.standard_normal(77))   # theta of a product state
    mats = [lowrank, zero, ident, graded, exp_decay, rank1, lowrank.T.copy()]
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])


def test_svd_qr_preconditioned_path(bb, rng):
    """Blocks with min(m, n) >= 48 go through QR preconditioning + deflation + completion from a full
    Householder Q: tall, wide, square, rank-deficient (theta = A.B has rank <= inner dimension), an
    all-zero block and a block whose leading columns are dependent (rank NOT revealed by unpivoted QR)."""
    tall = rng.standard_normal((700, 64))
    wide = rng.standard_normal((70, 500))
    square = rng.standard_normal((200, 200))
    theta_like = rng.standard_normal((300, 120)) @ rng.standard_normal((120, 260))
    theta_wide = rng.standard_normal((130, 50)) @ rng.standard_normal((50, 280))
    zero = np.zeros((96, 64))
    dep = rng.standard_normal((150, 100))
    dep[:, :40] = dep[:, 40:80] @ rng.standard_normal((40, 40))           # leading 40 columns depend on the next 40
    rank1 = np.outer(rng.standard_normal(90), rng.standard_normal(110))
    ident = np.eye(128)
    mats = [tall, wide, square, theta_like, theta_wide, zero, dep, rank1, ident]
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])


def test_svd_algorithm_names_and_errors(bb, rng):
    m = rng.standard_normal((12, 10))
    for algo in (None, 'gesdd', 'gesvd', 'robust', 'robust_silent', 'jacobi'):
        U, S, Vh = bb.matrix_svd(bb.as_block(m), algo)
        check_svd_invariants(m, bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh), TOL)
    with pytest.raises(ValueError):
        bb.matrix_svd(bb.as_block(m), 'no_such_driver')
    assert set(['gesdd', 'gesvd', 'robust', 'robust_silent']) <= set(bb.possible_svd_algorithms())
    assert bb.matrix_svd_batched([]) == []
    U, S, Vh = bb.matrix_svd(bb.zeros((0, 5)))
    assert U.shape == (0, 0) and S.shape == (0,) and Vh.shape == (0, 5)


def test_svd_of_strided_view(bb, rng):
    m = rng.standard_normal((40, 70))
    view = bb.permute_axes(bb.as_block(np.ascontiguousarray(m.T)), [1, 0])
    U, S, Vh = bb.matrix_svd(view)
    check_svd_invariants(m, bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh), TOL, sref=ops.matrix_svd(m)[1])


def test_svd_cfg2_block_sizes(bb, rng):
    """The SVD block list of BASELINE cfg2 (U(1) chi=1024: 13 blocks, largest 360^2)."""
    sizes = [8, 29, 74, 150, 246, 335, 360, 335, 246, 150, 74, 29, 8]
    mats = [rng.standard_normal((s, s)) for s in sizes]
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])


@pytest.mark.parametrize('full', [False, True])
def test_qr_lq(bb, rng, full):
    # the last five go through the blocked (GEMM-based) Householder path
    shapes = [(1, 1), (5, 3), (3, 5), (64, 64), (130, 40), (40, 130), (300, 17), (200, 150), (150, 200), (257, 257),
              (96, 96), (400, 129)]
    mats = [rng.standard_normal(s) for s in shapes]
    qrs = bb.matrix_qr_batched([bb.as_block(m) for m in mats], full)
    for m, (Q, R) in zip(mats, qrs):
        Q, R = bb.to_numpy(Q), bb.to_numpy(R)
        qref, rref = ops.matrix_qr(m, full)
        assert Q.shape == qref.shape and R.shape == rref.shape
        assert np.abs(Q @ R - m).max() <= TOL * np.abs(m).max()
        assert np.abs(Q.T @ Q - np.eye(Q.shape[1])).max() <= TOL
        assert np.abs(np.tril(R, -1)).max() == 0.0
        # same Householder sign convention as LAPACK dgeqrf: R agrees entry-wise
        assert np.abs(R - rref).max() <= TOL * np.abs(m).max() * max(m.shape)
    for m in mats[:5]:
        L, Q = bb.matrix_lq(bb.as_block(m), full)
        L, Q = bb.to_numpy(L), bb.to_numpy(Q)
        lref, qref = ops.matrix_lq(m, full)
        assert L.shape == lref.shape and Q.shape == qref.shape
        assert np.abs(L @ Q - m).max() <= TOL * np.abs(m).max()
        assert np.abs(Q @ Q.T - np.eye(Q.shape[0])).max() <= TOL


def test_qr_blocked_rank_deficient_and_zero_columns(bb, rng):
    m = rng.standard_normal((300, 60)) @ rng.standard_normal((60, 200))      # rank 60 < 200
    m[:, 7] = 0.0
    z = np.zeros((128, 100))
    for a in (m, z):
        for full in (False, True):
            Q, R = bb.matrix_qr(bb.as_block(a), full)
            Q, R = bb.to_numpy(Q), bb.to_numpy(R)
            assert np.abs(Q @ R - a).max() <= TOL * max(1.0, np.abs(a).max())
            assert np.abs(Q.T @ Q - np.eye(Q.shape[1])).max() <= TOL
            assert np.abs(np.tril(R, -1)).max() == 0.0


def test_qr_rank_deficient(bb, rng):
    m = rng.standard_normal((30, 4)) @ rng.standard_normal((4, 20))
    m[:, 3] = 0.0
    Q, R = bb.matrix_qr(bb.as_block(m), False)
    Q, R = bb.to_numpy(Q), bb.to_numpy(R)
    assert np.abs(Q @ R - m).max() <= TOL * np.abs(m).max()
    assert np.abs(Q.T @ Q - np.eye(20)).max() <= TOL


def test_eigh_batch_and_sort_options(bb, rng):
    mats = []
    for n in (1, 2, 7, 33, 64, 100, 191):
        a = rng.standard_normal((n, n))
        mats.append((a + a.T) / 2)
    mats.append(np.zeros((5, 5)))
    mats.append(np.array([[0.0, 1.0], [1.0, 0.0]]))                     # +-1: equal |lambda|, opposite sign
    mats.append(np.diag([3.0, 3.0, -3.0, 1e-9, 0.0]))                   # degenerate + tiny
    res = bb.eigh_batched([bb.as_block(m) for m in mats])
    for m, (W, V) in zip(mats, res):
        W, V = bb.to_numpy(W), bb.to_numpy(V)
        wref = ops.eigvalsh(m)
        scale = max(np.abs(wref).max(), 1e-300) if m.any() else 1.0
        assert np.all(np.diff(W) >= -TOL * scale)                       # ascending like np.linalg.eigh
        assert np.abs(W - wref).max() <= TOL * scale
        assert np.abs(m @ V - V * W).max() <= TOL * scale
        assert np.abs(V.T @ V - np.eye(len(W))).max() <= TOL
    h = mats[4]
    for sort in ('m>', 'm<', '>', '<', 'LM', 'SR'):
        W, V = bb.eigh(bb.as_block(h), sort)
        W, V = bb.to_numpy(W), bb.to_numpy(V)
        wref, _ = ops.eigh(h, sort)
        assert np.abs(W - wref).max() <= TOL * np.abs(wref).max()
        assert np.abs(h @ V - V * W).max() <= TOL * np.abs(wref).max()
    w = bb.to_numpy(bb.eigvalsh(bb.as_block(h)))
    assert np.abs(w - ops.eigvalsh(h)).max() <= TOL * np.abs(w).max()
    with pytest.raises(ValueError):
        bb.eigh(bb.as_block(h), 'bogus')
    with pytest.raises(ValueError):
        bb.eigh(bb.as_block(rng.standard_normal((3, 4))))


def _svd_check(bb, a, tol=1e-10):
    U, S, Vh = [bb.to_numpy(x) for x in bb.matrix_svd(bb.as_block(a))]
    scale = max(np.abs(a).max(), 1e-300)
    sref = np.linalg.svd(a / scale, compute_uv=False)
    assert np.isfinite(S).all()
    assert np.abs((U * S) @ Vh - a).max() <= tol * scale * max(a.shape)
    assert np.abs(S / scale - sref).max() <= tol * sref[0]
    assert np.abs(U.T @ U - np.eye(U.shape[1])).max() <= tol
    assert np.abs(Vh @ Vh.T - np.eye(Vh.shape[0])).max() <= tol


@pytest.mark.parametrize('n', [8, 40, 200])
def test_svd_hard_inputs(bb, rng, n):
    """Inputs on which LAPACK rescales or which are exactly rank deficient (regressions found with
    scripts/svd_hard_cases.py): entries near the ends of the double range, an all-ones block (its trailing
    blocks shrink by 1e-16 per elimination step into the denormal range; its zero rows tie with the padding
    rows of the engine), clustered singular values (linear, not quadratic, convergence at the end)."""
    g = rng.standard_normal((n, n))
    for sc in (1e150, 1e-150, 1e200, 1e-250):
        _svd_check(bb, sc * g)
    _svd_check(bb, np.ones((n, n)))
    _svd_check(bb, np.ones((n, n)) + np.outer(np.arange(n) == 0, np.arange(n) == 0))
    _svd_check(bb, np.ones((n + 5, n)))
    q1, _ = np.linalg.qr(rng.standard_normal((n, n)))
    q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
    _svd_check(bb, (q1 * np.r_[np.ones(n // 2), 1e-3 * np.ones(n - n // 2)]) @ q2)
    _svd_check(bb, (q1 * np.logspace(0, -15, n)) @ q2)


@pytest.mark.parametrize('shape', [(200, 200), (60, 40), (130, 250)])
def test_qr_eigh_hard_inputs(bb, rng, shape):
    m, n = shape
    mats = [np.ones((m, n)), 1e150 * rng.standard_normal((m, n)), 1e-200 * rng.standard_normal((m, n)),
            np.repeat(rng.standard_normal((m, max(n // 4, 1))), 4, axis=1)[:, :n]]
    for a in mats:
        Q, R = [bb.to_numpy(x) for x in bb.matrix_qr(bb.as_block(a), False)]
        scale = np.abs(a).max()
        assert np.abs(Q.T @ Q - np.eye(Q.shape[1])).max() <= 1e-10
        assert np.abs(Q @ R - a).max() <= 1e-10 * scale * max(m, n)
        assert np.abs(np.tril(R, -1)).max() == 0.0
    k = min(m, n)
    h = rng.standard_normal((k, k))
    h = h + h.T
    for sc in (1.0, 1e150, 1e-150, 1e250):
        w, v = [bb.to_numpy(x) for x in bb.eigh(bb.as_block(sc * h))]
        wr = np.linalg.eigvalsh(h)
        assert np.abs(w / sc - wr).max() <= 1e-10 * np.abs(wr).max()
        assert np.abs(v.T @ v - np.eye(k)).max() <= 1e-10


def test_svd_small_blocks_in_lds(bb, rng):
    """Blocks with min(m, n) <= 64 and max(m, n) <= 128 run the fused one-workgroup-per-block kernel (svd_small.hip):
    every shape class (tall, wide, odd column counts, single rows / columns), rank deficiency with Gram-Schmidt
    completion of the null directions, exact zeros, repeated columns, a 400-block batch as a non-abelian tensor has."""
    shapes = [(1, 1), (2, 1), (1, 2), (3, 3), (7, 5), (5, 7), (64, 64), (63, 64), (64, 63), (128, 64), (64, 128), (128, 1),
              (1, 128), (127, 33), (33, 127), (17, 17), (40, 40), (128, 63)]
    mats = [rng.standard_normal(s) for s in shapes]
    low = rng.standard_normal((50, 4)) @ rng.standard_normal((4, 33))
    dup = rng.standard_normal((30, 12))
    dup[:, 5] = dup[:, 2]
    dup[:, 9] = 0.0
    zero_rows = rng.standard_normal((40, 20))
    zero_rows[10:30] = 0.0
    mats += [low, low.T.copy(), dup, dup.T.copy(), zero_rows, np.zeros((9, 5)), np.zeros((5, 9)), np.eye(33), np.ones((20, 31)),
             np.diag(np.r_[np.ones(10), np.zeros(7)]), 1e-100 * rng.standard_normal((12, 12)), 1e100 * rng.standard_normal((12, 30)),
             1e-85 * rng.standard_normal((14, 9)), 1e85 * rng.standard_normal((9, 14))]   # inside the range the wrapper leaves alone
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    many = [rng.standard_normal((int(rng.integers(1, 41)), int(rng.integers(1, 41)))) for _ in range(400)]
    for m, (U, S, Vh) in zip(many, _svd_batch(bb, many)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    # strided views (a column slice of a larger block) and the mixed list small + QR-preconditioned
    big = rng.standard_normal((90, 150))
    B = bb.as_block(big)
    views = [bb.get_item(B, (slice(3, 60), slice(7, 47))), bb.permute_axes(bb.get_item(B, (slice(0, 30), slice(1, 100))), [1, 0]), B]
    refs = [big[3:60, 7:47], big[0:30, 1:100].T, big]
    for m, (u, s, vh) in zip(refs, bb.matrix_svd_batched(views)):
        check_svd_invariants(m, bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh), TOL, sref=ops.matrix_svd(m)[1])


@pytest.mark.parametrize('env', [{}, {'CYB_SVD_NOLQ': '1'}, {'CYB_SVD_LQ_FORCE_REDO': '1'}, {'CYB_SVD_NOMERGE': '1'},
                                 {'CYB_QR_FUSE': '1'}, {'CYB_QR_GEMM_UPDATE': '1'}, {'CYB_JACOBI_NOSWEEP': '1'}, {'CYB_QR_NOWAVE': '1'}, {'CYB_QR_NOMULTI': '1'}, {'CYB_JACOBI_ONELAUNCH': '1'}, {'CYB_JACOBI_PERSWEEP': '1'}, {'CYB_QR_NOSTOP': '1'}, {'CYB_QR_STOP_EVERY': '1'}, {'CYB_JACOBI_NODEFERJ': '1'},
                                 {'CYB_JACOBI_CROSS': '0'}, {'CYB_JACOBI_CROSS': '2'}, {'CYB_JACOBI_CROSS': '1000'}, {'CYB_QR_LA': '1'}, {'CYB_SVD_LQ_ALWAYS': '1'},
                                 {'CYB_SVD_LQ_SKIP_K': '2', 'CYB_SVD_LQ_SKIP_RATIO': '1e300'}, {'CYB_QR_CHUNK_ROWS': '0'}, {'CYB_QR_CHUNK_ROWS': '64'}],
                         ids=['default', 'no-lq', 'lq-fallback', 'no-merge', 'fuse', 'gemm-update', 'per-round', 'eight-wave-panels', 'one-workgroup-tall-panels', 'all-sweeps-in-one-launch', 'one-launch-per-sweep', 'no-early-stop', 'early-stop-test-every-step', 'j-update-not-deferred',
                              'all-rotation-sets-every-round', 'cross-only-every-second-round', 'cross-only-but-round-0', 'look-ahead-panel-workgroup', 'lq-for-every-block',
                              'no-lq-for-any-full-rank-block', 'whole-strips', 'strips-in-64-row-chunks'])
def test_svd_pipeline_variants(env):
    """Every switchable stage of the SVD pipeline against LAPACK on the same list: the default (QR -> LQ -> persistent
    block-Jacobi sweeps -> completion from Q2), the plain iteration on R (`CYB_SVD_NOLQ`), the FALLBACK from the LQ iteration
    to the plain one (forced with `CYB_SVD_LQ_FORCE_REDO`: in production it is taken when a row of S Z^T is exactly zero or
    the iteration does not settle), small blocks in an iteration of their own, the panel factorisation inside the strip launch
    (opt-in), the grouped-GEMM form of the block-reflector application, one launch per Jacobi round, all sweeps in ONE launch
    with the convergence test on the device (the default for lists of small matrices only) / one launch per sweep for every list,
    the first QR without / with a per-step test of its early stop.  The switches are read
    once per process, hence the child process (one GPU process at a time)."""
    import os
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), 'svd_env_worker.py')], env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('OK'), r.stdout[-2000:] + r.stderr[-4000:]


def test_svd_first_qr_stops_at_the_numerical_rank(bb, rng):
    """The first QR of the pipeline stops factoring once the trailing block is at the rounding level of the matrix
    (blocked_qr.hip, `panel_stop_check`; a theta = A.B has half the rank of its extents).  Ranks on and around the 32-column
    panel boundaries and the every-other-step measurement, rank 1, a rank-deficient block whose first columns are 1e8 times
    larger than the rest (the reference norm the kernels use underestimates ||A||: they must stop later, not wrongly), and
    full-rank / graded blocks that must not stop: LAPACK's values, reconstruction, isometries -- null vectors included."""
    mats = []
    for m, n, r in [(200, 180, 31), (200, 180, 32), (200, 180, 33), (300, 300, 64), (300, 300, 65), (300, 300, 95), (300, 300, 97),
                    (500, 260, 1), (260, 500, 130), (700, 700, 350), (1500, 400, 200)]:
        mats.append(rng.standard_normal((m, r)) @ rng.standard_normal((r, n)))
    big_first = rng.standard_normal((400, 100)) @ rng.standard_normal((100, 300))
    big_first[:, :40] *= 1e8
    q1, _ = np.linalg.qr(rng.standard_normal((300, 300)))
    q2, _ = np.linalg.qr(rng.standard_normal((300, 300)))
    mats += [big_first, (q1 * np.logspace(0, -15, 300)) @ q2, rng.standard_normal((330, 330)),
             rng.standard_normal((320, 40)) @ rng.standard_normal((40, 320)) + 1e-9 * rng.standard_normal((320, 320))]
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    # panels of more than 1536 rows (multi-workgroup panel kernel) stop the same way
    tall = [rng.standard_normal((2000, 300)) @ rng.standard_normal((300, 700)), rng.standard_normal((1800, 90)) @ rng.standard_normal((90, 1800))]
    for m, (U, S, Vh) in zip(tall, _svd_batch(bb, tall)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    # the truncating caller's form reports the numerical ranks (numpy.linalg.matrix_rank's threshold)
    low = mats[:7]
    _, ranks = bb.matrix_svd_batched([bb.as_block(m) for m in low], null_vectors=False, return_rank=True)
    assert list(ranks) == [31, 32, 33, 64, 65, 95, 97]


def test_svd_eightfold_singular_values_keep_their_vectors_orthogonal(bb, rng):
    """Blocks whose singular values come in groups of eight equal ones (found by `scripts/svd_fuzz.py`): rows of equal norm
    used to be rotated by 45 degrees on couplings of pure rounding noise, which made the convergence linear, and the
    prediction of the last sweep then left 3e-10 ... 8e-10 of non-orthogonality between vectors of DIFFERENT values behind.
    The pivot eigensolve now leaves couplings below tol / 2 alone (`jacobi_rot_bf`).  Tall, wide and square, alone and
    inside a list, real and complex."""
    mats = []
    for m, n in [(285, 67), (18, 566), (566, 18), (200, 200), (96, 410), (330, 330)]:
        k = min(m, n)
        q1, _ = np.linalg.qr(rng.standard_normal((m, k)))
        q2, _ = np.linalg.qr(rng.standard_normal((n, k)))
        mats.append((q1 * np.repeat(rng.random(k // 8 + 1) + 0.1, 8)[:k]) @ q2.T)
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    for m in mats[:3]:
        (U, S, Vh), = _svd_batch(bb, [m])
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    zs = []
    for m, n in [(150, 120), (64, 300)]:
        k = min(m, n)
        q1, _ = np.linalg.qr(rng.standard_normal((m, k)) + 1j * rng.standard_normal((m, k)))
        q2, _ = np.linalg.qr(rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k)))
        zs.append((q1 * np.repeat(rng.random(k // 8 + 1) + 0.1, 8)[:k]) @ q2.conj().T)
    for z, (u, s, vh) in zip(zs, bb.matrix_svd_batched([bb.as_block(z) for z in zs])):
        u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
        k = min(z.shape)
        assert np.abs((u * s) @ vh - z).max() <= TOL * np.linalg.norm(z) and np.abs(s - np.linalg.svd(z, compute_uv=False)).max() <= TOL
        assert np.abs(u.conj().T @ u - np.eye(k)).max() <= TOL and np.abs(vh @ vh.conj().T - np.eye(k)).max() <= TOL


def test_eigh_of_a_gram_matrix_with_a_large_null_space(bb, rng):
    """A 260 x 260 Gram matrix of rank 132 (found by `scripts/svd_fuzz.py`, seed 301; `tests/golden/eigh_fuzz_seed301_list25_260.npz`):
    its 128-fold zero eigenvalue is a cluster of equal-norm rows in the shifted iteration, the off-norm went 5.0e-7 -> 3.2e-10 ->
    1.6e-10 -> 1.3e-14, and the predicted stop after the second of these left 1.6e-10 of non-orthogonality between a null
    vector and a range vector.  The prediction now needs a quadratic step behind it (`kPredictQuad`).  The fixture and fresh
    Gram matrices of the same kind, alone and inside a list; errors relative to n * max|h| as in the soak."""
    hs = [np.load(os.path.join(os.path.dirname(__file__), 'golden', 'eigh_fuzz_seed301_list25_260.npz'))['arr_0']]
    for n, r in [(260, 132), (200, 40), (333, 300), (128, 1)]:
        b = rng.standard_normal((n, r)) * np.logspace(0, -2, r)
        hs.append(b @ b.T)
    for lst in ([hs[0]], hs):
        for h, (w, v) in zip(lst, bb.eigh_batched([bb.as_block(h) for h in lst])):
            w, v = bb.to_numpy(w), bb.to_numpy(v)
            nrm = np.abs(h).max() * h.shape[0]
            assert np.abs(w - np.linalg.eigvalsh(h)).max() <= 1e-12 * nrm
            assert np.abs(h @ v - v * w).max() <= 1e-12 * nrm
            assert np.abs(v.T @ v - np.eye(h.shape[0])).max() <= 1e-11


def test_svd_rank_deficient_blocks_with_many_zero_columns(bb, rng):
    """Two sector blocks of a composed tensor (38 x 115 of rank 2, 38 x 135 of rank 19 with 105 zero columns; found by
    `scripts/tensor_fuzz.py`, seed 11, kept as tests/golden/svd_fuzz_seed11_round305.npz): too wide for the in-LDS kernel and
    too small for the pipeline's old lower limit, the second one went to the direct iteration, which has no deflation of
    numerically zero rows and did not converge in 40 sweeps.  Every block outside the in-LDS kernel now takes the pipeline.
    Also: synthetic blocks of the same kind, both orientations, alone and in a list."""
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'svd_fuzz_seed11_round305.npz'))
    mats = [d[k] for k in d.files]
    for m, n, r, nzero in [(38, 135, 19, 105), (135, 38, 19, 0), (20, 300, 7, 200), (47, 129, 47, 60), (2, 400, 1, 300), (40, 160, 0, 160)]:
        a = rng.standard_normal((m, r)) @ rng.standard_normal((r, n)) if r else np.zeros((m, n))
        a[:, rng.permutation(n)[:nzero]] = 0.0
        mats.append(a)
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    for m in mats:
        (U, S, Vh), = _svd_batch(bb, [m])
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    # the same blocks with complex entries (same zero columns), in a list and alone
    # (a phase per row and per column keeps ranks and zero columns)
    zs = [np.exp(2j * np.pi * rng.random((m.shape[0], 1))) * m * np.exp(2j * np.pi * rng.random((1, m.shape[1]))) for m in mats]
    got = bb.matrix_svd_batched([bb.as_block(z) for z in zs]) + [bb.matrix_svd_batched([bb.as_block(z)])[0] for z in zs]
    for z, (u, s, vh) in zip(zs + zs, got):
        u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
        k, nrm = min(z.shape), max(np.linalg.norm(z), 1e-300)
        assert np.abs((u * s) @ vh - z).max() <= TOL * nrm and np.abs(s - np.linalg.svd(z, compute_uv=False)).max() <= TOL * nrm
        assert np.abs(u.conj().T @ u - np.eye(k)).max() <= TOL and np.abs(vh @ vh.conj().T - np.eye(k)).max() <= TOL


def test_svd_rank_deficiency_hidden_from_the_row_norms_of_r(bb, rng):
    """A 462 x 600 block with 140 zero rows (found by `scripts/svd_fuzz.py`, seed 53): its transpose has 140 zero COLUMNS, the R
    of the first QR has zero columns but no small row, so nothing is deflated up front and the LQ iteration ends with 140 rows
    of rounding noise whose normalised directions were taken for left vectors: |Vh Vh^T - 1| = 0.8.  Such rows now send the
    list to the plain iteration (deflation on, completion by QR).  Zero rows / zero columns / a product of block-sparse
    factors, both orientations, alone and in a list."""
    mats = []
    for m, n in [(462, 600), (600, 462), (300, 300), (150, 700)]:
        a = rng.standard_normal((m, n))
        a[rng.random(m) < 0.3] = 0.0
        mats.append(a)
        a = rng.standard_normal((m, n))
        a[:, rng.random(n) < 0.3] = 0.0
        mats.append(a)
        r = min(m, n) // 2
        b1, b2 = rng.standard_normal((m, r)), rng.standard_normal((r, n))
        b1[rng.random((m, 1)) < 0.5 * np.ones((1, r)) * (np.arange(r) % 2)] = 0.0
        b2[:, rng.random(n) < 0.4] = 0.0
        b2[np.arange(r) % 3 == 0, : n // 2] = 0.0
        mats.append(b1 @ b2)
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    for m in mats[:3]:
        (U, S, Vh), = _svd_batch(bb, [m])
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])


def test_svd_dmrg_theta_sectors_converge_in_few_sweeps(bb):
    """The coupled-charge sectors of a two-site theta from the toy DMRG (Heisenberg L=32, chi=256, centre bond; dumped
    from tests/toy_dmrg.py on the device into tests/golden/dmrg_theta_chi256_center.npz): singular spectra graded over 14
    decades, the matrices the path exists for.  LAPACK's values, reconstruction and isometries to 1e-10 -- and the
    block-Jacobi iteration must settle within 9 sweeps: on the rows of R it took 13-15, the second (LQ)
    preconditioning step brought it to 6-8."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'dmrg_theta_chi256_center.npz'))
    mats = [np.ascontiguousarray(d[k]) for k in sorted(d.files, key=lambda s: int(s[1:]))]
    res, info = bb.matrix_svd_batched([bb.as_block(m) for m in mats], return_info=True)
    for m, (U, S, Vh) in zip(mats, res):
        check_svd_invariants(m, bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh), TOL, sref=ops.matrix_svd(m)[1])
    assert max(int(i) for i in info) <= 9, list(info)


def test_svd_and_eigh_lists_with_more_pairs_than_workgroups(bb, rng):
    """A list whose block pairs outnumber the resident workgroups of the persistent sweep kernel (256 CUs, one workgroup
    each): 20 full-rank 420 x 420 blocks are 280 pairs per round, so the short matrices share workgroups that run several
    entries one after the other (jacobi_engine.hip, `wgent`); same for a hermitian list through `eigh`."""
    mats = [rng.standard_normal((420, 420)) for _ in range(17)] + [rng.standard_normal((600, 600)) for _ in range(3)]
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
    herm = [(lambda x: x + x.T)(rng.standard_normal((n, n))) for n in [430] * 16 + [640] * 3]
    res = bb.eigh_batched([bb.as_block(h) for h in herm]) if hasattr(bb, 'eigh_batched') else [bb.eigh(bb.as_block(h)) for h in herm]
    for h, (w, v) in zip(herm, res):
        w, v = bb.to_numpy(w), bb.to_numpy(v)
        nrm = np.abs(h).max() * h.shape[0]
        assert np.abs(np.sort(w) - np.linalg.eigvalsh(h)).max() <= TOL * nrm
        assert np.abs(h @ v - v * w).max() <= TOL * nrm
        assert np.abs(v.T @ v - np.eye(len(w))).max() <= TOL


def test_qr_and_svd_of_blocks_with_more_than_1536_rows(bb, rng):
    """Panels of more than 1536 rows are factored by several workgroups that exchange their partial column dots once per
    column step (`qr_panel_multi_kernel`): tall, very tall, wide and square blocks, QR against its defining properties and
    the SVD against LAPACK."""
    mats = [rng.standard_normal(s) for s in [(1700, 300), (3100, 40), (300, 1700), (4700, 33), (1600, 1600)]]
    for a, (q, r) in zip(mats, bb.matrix_qr_batched([bb.as_block(m) for m in mats])):
        q, r = bb.to_numpy(q), bb.to_numpy(r)
        k = min(a.shape)
        assert np.abs(q @ r - a).max() <= TOL * np.linalg.norm(a)
        assert np.abs(q.T @ q - np.eye(k)).max() <= TOL and np.abs(np.tril(r, -1)).max() == 0.0
    for m, (U, S, Vh) in zip(mats, _svd_batch(bb, mats)):
        check_svd_invariants(m, U, S, Vh, TOL, sref=ops.matrix_svd(m)[1])
