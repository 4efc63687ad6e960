"""Per-block reference operations: the numpy/scipy calls NumpyBlockBackend makes.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each function cites the reference call site.
"""
import numpy as np
import scipy.linalg


def matrix_dot(a, b):
    """src/block_backend/numpy.cpp:1218-1225 -- ``np.dot(a, b)``."""
    return np.dot(a, b)


def tdot(a, b, idcs_a, idcs_b):
    """numpy.cpp:1118-1129 -- ``np.tensordot(a, b, (idcs_a, idcs_b))``."""
    return np.tensordot(a, b, (list(idcs_a), list(idcs_b)))


def matrix_svd(a, algorithm=None):
    """numpy.cpp:1247-1297 -- ``scipy.linalg.svd(a, full_matrices=False)`` (gesdd; gesvd on request,
    'robust' retries with gesvd when gesdd raises)."""
    algo = algorithm or 'gesdd'
    if algo == 'gesdd':
        return scipy.linalg.svd(a, full_matrices=False)
    if algo == 'gesvd':
        return scipy.linalg.svd(a, full_matrices=False, lapack_driver='gesvd')
    if algo in ('robust', 'robust_silent'):
        try:
            return scipy.linalg.svd(a, full_matrices=False)
        except np.linalg.LinAlgError:
            if algo != 'robust_silent':
                raise
        return scipy.linalg.svd(a, full_matrices=False, lapack_driver='gesvd')
    raise ValueError('SVD algorithm not supported: ' + str(algo))


def matrix_qr(a, full):
    """numpy.cpp:1236-1245 -- ``scipy.linalg.qr(a, mode='full' if full else 'economic')``."""
    return scipy.linalg.qr(a, mode='full' if full else 'economic')


def matrix_lq(a, full):
    """src/block_backend/block_backend.cpp:1033-1040 -- q, r = qr(a.T); return r.T, q.T."""
    q, r = matrix_qr(a.T, full)
    return r.T, q.T


def argsort(w, sort):
    """block_backend.cpp:759-781."""
    if sort in ('m<', 'SM'):
        key = np.abs(w)
    elif sort in ('m>', 'LM'):
        key = -np.abs(w)
    elif sort in ('<', 'SR', 'SA'):
        key = np.real(w)
    elif sort in ('>', 'LR', 'LA'):
        key = -np.real(w)
    else:
        raise ValueError(f"Unknown sort option: '{sort}'")
    return np.argsort(key, kind='stable')


def eigh(a, sort=None):
    """numpy.cpp:658-680 -- ``np.linalg.eigh`` (+ optional re-sort of w and the columns of v)."""
    w, v = np.linalg.eigh(a)
    if sort is not None:
        perm = argsort(w, sort)
        w, v = np.take(w, perm), np.take(v, perm, axis=1)
    return w, v


def eigvalsh(a, sort=None):
    """numpy.cpp:682-698."""
    w = np.linalg.eigvalsh(a)
    if sort is not None:
        w = np.take(w, argsort(w, sort))
    return w


def norm(a):
    """numpy.cpp:898-913 -- ``np.linalg.norm(a.ravel())``."""
    return float(np.linalg.norm(np.asarray(a).ravel()))


def inner(a, b, do_dagger):
    """numpy.cpp:815-842."""
    if do_dagger:
        return float(np.tensordot(np.conj(a), b, a.ndim))
    return float(np.tensordot(a, b, [list(range(a.ndim)), list(reversed(range(a.ndim)))]))


def scale_axis(block, factors, axis):
    """numpy.cpp:1373-1385."""
    idx = [None] * block.ndim
    idx[axis] = slice(None)
    return block * factors[tuple(idx)]


def apply_mask(block, mask, ax):
    """numpy.cpp:605-613 -- ``np.compress(mask, block, ax)``."""
    return np.compress(mask, block, ax)


def enlarge_leg(block, mask, axis):
    """numpy.cpp:700-728."""
    shape = list(block.shape)
    shape[axis] = len(mask)
    res = np.zeros(shape, dtype=block.dtype)
    idx = [slice(None)] * block.ndim
    idx[axis] = mask
    res[tuple(idx)] = block
    return res


def combine_legs(a, leg_idcs_combine, cstyles=True):
    """block_backend.cpp:784-829."""
    if isinstance(cstyles, bool):
        cstyles = [cstyles] * len(leg_idcs_combine)
    perm, shape, k = [], [], 0
    groups = {g[0]: (g, c) for g, c in zip(leg_idcs_combine, cstyles)}
    member = {i for g in leg_idcs_combine for i in g}
    while k < a.ndim:
        if k in groups:
            g, c = groups[k]
            g = list(g) if c else list(reversed(g))
            perm += g
            shape.append(int(np.prod([a.shape[i] for i in g])))
            k = max(g) + 1
        elif k in member:
            k += 1
        else:
            perm.append(k)
            shape.append(a.shape[k])
            k += 1
    return np.reshape(np.transpose(a, perm), shape)
