"""A numpy stand-in with the grouped block-backend interface, for CPU tests of the HOST logic in
``cyten_amd.abelian`` / ``cyten_amd.sharding`` (test infrastructure; built on the oracle ops)."""
import numpy as np

from oracle import block_ops as ops


class _Plan:
    def __init__(self, groups):
        self.groups = groups
        self.flops = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for g in groups for a, b in g)

    def run(self):
        outs = []
        for g in self.groups:
            acc = ops.matrix_dot(*g[0])
            for a, b in g[1:]:
                acc = acc + ops.matrix_dot(a, b)
            outs.append(acc)
        return outs

    def destroy(self):
        pass


class NumpyGroupedBackend:
    def as_block(self, a, dtype=None, device=None):
        return np.array(a, dtype=float)

    def to_numpy(self, a):
        return np.asarray(a)

    def zeros(self, shape, dtype=None, device=None):
        return np.zeros(shape)

    def zeros_many(self, shapes, dtype=None, device=None):
        return [np.zeros(sh, dtype=dtype or float) for sh in shapes]

    def get_item(self, a, key):
        return a[key]

    def reshape(self, a, shape):
        return np.reshape(a, shape)

    def permute_axes(self, a, perm):
        return np.transpose(a, perm)

    def contiguous(self, a):
        return np.ascontiguousarray(a)

    def contiguous_many(self, blocks):
        return list(blocks)

    def copy_many(self, pairs):
        for d, s in pairs:
            d[...] = s

    def make_gemm_plan(self, groups, outs=None):
        return _Plan(groups)

    def matrix_dot_grouped(self, groups):
        return _Plan(groups).run()

    def matrix_svd_batched(self, blocks, algorithm=None):
        return [ops.matrix_svd(b, None if algorithm == 'jacobi' else algorithm) for b in blocks]

    def matrix_qr_batched(self, blocks, full=False):
        return [ops.matrix_qr(b, full) for b in blocks]

    def eigh_batched(self, blocks, sort=None):
        return [ops.eigh(b, sort) for b in blocks]

    def mask_gather_many(self, items):
        return [ops.apply_mask(a, np.asarray(m), ax) for a, m, ax in items]

    def enlarge_leg_many(self, items):
        return [ops.enlarge_leg(a, np.asarray(m), ax) for a, m, ax in items]

    def matrix_lq_batched(self, blocks, full=False):
        return [ops.matrix_lq(b, full) for b in blocks]

    def eye_matrix(self, dim, dtype=None, device=None):
        return np.eye(dim)

    def is_correct_block_type(self, b):
        return isinstance(b, np.ndarray)

    def to_dtype(self, a, dtype):
        return np.asarray(a, dtype=dtype)

    def as_device(self, device):
        return 'cpu'

    def norm_many(self, blocks):
        return float(np.sqrt(sum(np.sum(np.square(b)) for b in blocks)))

    def inner_many(self, xs, ys):
        return float(sum(np.sum(x * y) for x, y in zip(xs, ys)))

    def linear_combination_many(self, a_coef, vs, b_coef, ws):
        return [a_coef * v + b_coef * w for v, w in zip(vs, ws)]

    def mul_many(self, a, blocks):
        return [a * b for b in blocks]

    def mul(self, a, block):
        return a * block

    def scale_axis(self, block, factors, axis):
        return ops.scale_axis(block, np.asarray(factors), axis)

    def conj(self, a):
        return np.conj(a)
