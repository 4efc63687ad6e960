"""The Array-API namespace that carries an unmodified cyten onto the device (integration/hip_array_api.py), function by
function against numpy, in the call patterns of the reference's ArrayApiBlockBackend (src/block_backend/array_api.cpp)."""
import numpy as np
import pytest

from integration.hip_array_api import HipArray, HipArrayNamespace

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope='module')
def xp():
    return HipArrayNamespace('cuda:0')


def close(got, want, tol=TOL):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.abs(got - want).max(initial=0.0) <= tol * max(1.0, np.abs(want).max(initial=0.0))


def test_creation_dtypes_and_views(xp, rng):
    a = rng.standard_normal((4, 5, 3))
    x = xp.asarray(a)
    assert isinstance(x, HipArray) and x.shape == (4, 5, 3) and x.dtype == np.float64 and x.device == 'cuda:0'
    close(1.0 * x, a)                                                     # py_mul(1.0, block) of as_block (array_api.cpp:595)
    assert xp.zeros((2, 3), dtype=xp.float64).shape == (2, 3) and float(np.asarray(xp.sum(xp.zeros((2, 3))))) == 0.0
    close(xp.ones((3,), dtype=xp.float64, device='cuda:0'), np.ones(3))   # as_device probe (array_api.cpp:604)
    close(xp.eye(4, dtype=xp.complex128), np.eye(4))
    assert xp.asarray(3, dtype=xp.int64).dtype == np.int64 and xp.asarray(np.float32(2.0), dtype=xp.float32).dtype == np.float32
    assert xp.asarray(True, dtype=xp.bool).dtype == np.bool_
    close(xp.reshape(x, (20, 3)), a.reshape(20, 3))
    close(xp.permute_dims(x, (2, 0, 1)), a.transpose(2, 0, 1))
    close(xp.expand_dims(x, axis=1), a[:, None])
    close(xp.squeeze(xp.expand_dims(x, axis=0), 0), a)
    close(xp.astype(x, xp.complex128), a.astype(complex))
    close(xp.concat([x, x], axis=1), np.concatenate([a, a], axis=1))
    with pytest.raises(ValueError):
        xp.zeros((2,), device='cpu')


def test_elementwise_comparisons_and_scalars(xp, rng):
    a, b = rng.standard_normal((6, 7)), rng.standard_normal((6, 7))
    x, y = xp.asarray(a), xp.asarray(b)
    for got, want in [(x + y, a + b), (x - y, a - b), (x * y, a * b), (x / y, a / b), (2.5 * x, 2.5 * a), (x * 2.5, 2.5 * a),
                      (1.0 - x, 1.0 - a), (x - 1.0, a - 1.0), (1.0 / y, 1.0 / b), (x ** 2, a ** 2), (abs(x), np.abs(a))]:
        close(got, want)
    for got, want in [(x < y, a < b), (x <= y, a <= b), (x > y, a > b), (x >= y, a >= b), (x == x, a == a), (x != y, a != b), (x < 0.0, a < 0.0)]:
        np.testing.assert_array_equal(np.asarray(got), want)
    z = a + 1j * b
    w = xp.asarray(z)
    close(xp.conj(w), z.conj())
    close(xp.real(w), a)
    close(xp.imag(w), b)
    close(xp.abs(w), np.abs(z))
    close(xp.exp(x), np.exp(a))
    close(xp.log(xp.abs(x)), np.log(np.abs(a)))
    # cutoff_inverse (array_api.cpp:720-727): 1 / where(|a| < cutoff, inf, a)
    denom = xp.where(xp.abs(x) < 0.3, xp.asarray(np.inf), x)
    close(1.0 / denom, 1.0 / np.where(np.abs(a) < 0.3, np.inf, a))
    close(xp.where(x < y, x, y), np.minimum(a, b))
    assert np.asarray(xp.all(x == x)).item() is True and np.asarray(xp.any(x != x)).item() is False
    assert xp.asarray(2.5).item() == 2.5 and xp.asarray(1 + 2j).item() == 1 + 2j


def test_indexing_patterns_of_the_reference(xp, rng):
    a = rng.standard_normal((5, 6))
    x = xp.asarray(a)
    w = rng.standard_normal(6)
    perm = xp.argsort(xp.asarray(w), axis=0)
    np.testing.assert_array_equal(np.asarray(perm), np.argsort(w, kind='stable'))
    close(xp.asarray(w)[perm], w[np.argsort(w)])                            # eigh(sort): w[perm]          (array_api.cpp:737)
    close(x[(slice(None), perm)], a[:, np.argsort(w)])                      # ... and v[:, perm]           (:738-740)
    close(x[(2, 3)], a[2, 3])
    mask = xp.asarray(np.array([True, False, True, True, False]))
    close(x[(mask, slice(None))], a[[0, 2, 3]])                             # apply_mask (:697)
    big = xp.zeros((5, 6), dtype=xp.float64)
    big[(mask, slice(None))] = xp.asarray(a[[0, 2, 3]])                     # enlarge_leg (:755-766)
    want = np.zeros((5, 6))
    want[[0, 2, 3]] = a[[0, 2, 3]]
    close(big, want)
    y = xp.asarray(a.copy())
    y[(slice(1, 3), slice(0, 2))] = xp.asarray(np.ones((2, 2)))
    want = a.copy()
    want[1:3, 0:2] = 1.0
    close(y, want)
    assert np.asarray(xp.argmax(xp.abs(x))).item() == np.argmax(np.abs(a)) and np.asarray(xp.argmin(x)).item() == np.argmin(a)
    close(xp.diagonal(xp.asarray(a[:5, :5])), np.diag(a[:5, :5]))


def test_reductions_and_linalg(xp, rng):
    a = rng.standard_normal((7, 5))
    x = xp.asarray(a)
    close(xp.sum(x), a.sum())
    close(xp.sum(x, axis=0), a.sum(axis=0))
    close(xp.max(x), a.max())
    close(xp.min(x), a.min())
    close(xp.linalg.vector_norm(x), np.linalg.norm(a))
    t = rng.standard_normal((3, 4, 4))
    close(xp.linalg.trace(xp.asarray(t)), np.trace(t, axis1=-2, axis2=-1))
    close(xp.linalg.trace(xp.asarray(t[0])), np.trace(t[0]))
    b = rng.standard_normal((5, 6))
    close(xp.matmul(x, xp.asarray(b)), a @ b)
    c = rng.standard_normal((4, 5, 6))
    close(xp.tensordot(xp.asarray(c), xp.asarray(b), ([1, 2], [0, 1])), np.tensordot(c, b, ([1, 2], [0, 1])))
    u, s, vh = xp.linalg.svd(x, full_matrices=False)
    close(s, np.linalg.svd(a, compute_uv=False))
    close(np.asarray(u) * np.asarray(s) @ np.asarray(vh), a)
    q, r = xp.linalg.qr(x)
    close(np.asarray(q) @ np.asarray(r), a)
    assert np.abs(np.tril(np.asarray(r), -1)).max() == 0.0
    h = a.T @ a
    w, v = xp.linalg.eigh(xp.asarray(h))
    close(w, np.linalg.eigvalsh(h))
    close(np.asarray(v) * np.asarray(w) @ np.asarray(v).T, h)
    close(xp.linalg.eigvalsh(xp.asarray(h)), np.linalg.eigvalsh(h))


def test_contraction_loop_is_deferred_into_one_launch(xp, rng):
    """the per-pair matmul + `+` chain of abelian_compose_worker (abelian.cpp:1437-1446) through the namespace"""
    bb = xp.bb
    bb.flush()
    f0 = bb.n_flushes
    A = [rng.standard_normal((20, 8 + k)) for k in range(4)]
    B = [rng.standard_normal((8 + k, 30)) for k in range(4)]
    outs = []
    for rep in range(3):
        blk = xp.matmul(xp.asarray(A[0]), xp.asarray(B[0]))
        for k in range(1, 4):
            blk = blk + xp.matmul(xp.asarray(A[k]), xp.asarray(B[k]))
        outs.append(xp.reshape(blk, (4, 5, 30)))
    assert bb.n_flushes == f0                                     # nothing launched yet
    want = sum(a @ b for a, b in zip(A, B)).reshape(4, 5, 30)
    for o in outs:
        close(o, want)
    assert bb.n_flushes == f0 + 1                                 # ONE grouped launch for all twelve products


class _StandInCore:
    """What integration/cyten_hip.py needs from ``cyten._core`` for its overrides, restated from the reference's text:
    ``ArrayApiBlockBackend::Block::to_numpy`` = ``numpy.asarray(arr_)`` (array_api.cpp:158-161), ``as_block`` /
    ``block_from_numpy`` = ``api.asarray(obj, dtype=, device=)`` wrapped into a block (array_api.cpp:568-600, 792-802).
    cyten itself is not importable on the GPU box, so the overrides are exercised against this stand-in."""

    class Block:
        def __init__(self, arr):
            self.arr = arr

        def to_numpy(self):
            return np.asarray(self.arr)

    class _Dtype:
        def __init__(self, dt):
            self.dt = np.dtype(dt)

        def to_numpy_dtype(self):
            return self.dt

    class ArrayApiBlockBackend:
        def __init__(self, namespace, default_device):
            self.api = namespace

        def as_block(self, a, dtype=None, device=None):
            return _StandInCore.Block(self.api.asarray(a))

        def block_from_numpy(self, a, dtype=None, device=None):
            return _StandInCore.Block(self.api.asarray(a))


def test_adapter_overrides_run_on_the_device(xp, rng, monkeypatch):
    """The eight operations the C++ ArrayApiBlockBackend leaves to the Python subclass (integration/cyten_hip.py): against
    numpy / scipy, and WITHOUT a host copy of the operands (Block::to_numpy under the namespace's passthrough hands over the
    device array itself; only block_from_mask reads its boolean vector)."""
    import scipy.linalg
    from integration import cyten_hip
    be = cyten_hip._make_backend_class(_StandInCore)(xp, xp.device)
    blk = lambda a: be.as_block(xp.asarray(a))
    a, b = rng.standard_normal((5, 4)), rng.standard_normal((3, 2))
    z = a + 1j * rng.standard_normal((5, 4))
    h = rng.standard_normal((6, 6))
    pos = np.abs(a) + 0.1
    d = rng.standard_normal(7)
    operands = [blk(v) for v in (z, pos, d, h, a, b, z.real + 0j)]
    n_d2h = [0]
    real_to_numpy = xp.bb.to_numpy
    monkeypatch.setattr(xp.bb, 'to_numpy', lambda *args, **kw: (n_d2h.__setitem__(0, n_d2h[0] + 1), real_to_numpy(*args, **kw))[1])
    got = [be.angle(operands[0]), be.sqrt(operands[1]), be.block_from_diagonal(operands[2]), be.matrix_exp(operands[3]),
           be.kron(operands[4], operands[5]), be.tile(operands[2], 3), be.real_if_close(operands[6], 100.0), be.real_if_close(operands[0], 100.0)]
    assert n_d2h[0] == 0, 'an override went through the host'
    monkeypatch.undo()
    want = [np.angle(z), np.sqrt(pos), np.diag(d), scipy.linalg.expm(h), np.kron(a, b), np.tile(d, 3), z.real, z]
    for g, w in zip(got, want):
        assert isinstance(g.arr, HipArray) and g.arr.blk is not None      # the result is a device array
        close(g.arr, w, 1e-12)
    assert got[6].arr.dtype == np.float64 and got[7].arr.dtype == np.complex128
    mask = rng.random(9) < 0.5
    m = be.block_from_mask(blk(mask), _StandInCore._Dtype(np.float64))
    want_m = np.zeros((int(mask.sum()), 9))
    want_m[np.arange(int(mask.sum())), np.flatnonzero(mask)] = 1
    np.testing.assert_array_equal(np.asarray(m.arr), want_m)          # (N, M) as numpy.cpp:748-766
    # host-held arrays (index data) keep the reference's numpy pattern
    idx = be.as_block(xp.asarray(np.arange(4), dtype=xp.int64))
    np.testing.assert_array_equal(np.asarray(be.tile(idx, 2).arr), np.tile(np.arange(4), 2))
    # outside the passthrough, numpy.asarray is a host copy as before
    assert np.asarray(operands[1].arr).dtype == np.float64
