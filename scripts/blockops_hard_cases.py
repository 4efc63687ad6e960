"""Probe of the block-list data movement / BLAS-1 kernels on awkward shapes (vs numpy)."""
import sys, itertools
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(3)
bad = 0
def eq(name, got, want, tol=0.0):
    global bad
    ok = got.shape == want.shape and (np.abs(got - want).max() <= tol if want.size else True)
    if not ok:
        bad += 1
        print('MISMATCH', name, got.shape, want.shape, np.abs(got - want).max() if got.shape == want.shape and want.size else '')
# permutations of up to 8 axes, with singleton and odd extents
for shape in [(3, 1, 4, 2, 1, 5), (2, 3, 2, 2, 3, 2, 2, 2), (7,), (1, 1, 1), (5, 1), (129, 3, 65), (64, 64), (33, 1, 31, 2)]:
    a = rng.standard_normal(shape)
    x = bb.as_block(a)
    for _ in range(4):
        perm = list(rng.permutation(len(shape)))
        eq(f'permute {shape} {perm}', bb.to_numpy(bb.contiguous(bb.permute_axes(x, perm))), np.ascontiguousarray(a.transpose(perm)))
    eq('reshape flat', bb.to_numpy(bb.reshape(x, (-1,))), a.reshape(-1))
# empty shapes
for shape in [(0,), (0, 5), (4, 0, 3)]:
    a = np.zeros(shape)
    eq(f'empty {shape}', bb.to_numpy(bb.as_block(a)), a)
    eq(f'empty permute {shape}', bb.to_numpy(bb.contiguous(bb.permute_axes(bb.as_block(a), list(range(len(shape)))[::-1]))), a.transpose())
# slicing + set_item
a = rng.standard_normal((9, 11, 7)); x = bb.as_block(a)
for key in [(slice(2, 7), slice(None), slice(1, 6, 2)), (3, slice(None), slice(None)), (slice(None), 4, 2), (slice(0, 0), slice(None), slice(None))]:
    eq(f'get_item {key}', bb.to_numpy(bb.get_item(x, key)), a[key])
b = rng.standard_normal((3, 11, 2)); y = bb.as_block(a.copy()); bb.set_item(y, (slice(1, 4), slice(None), slice(2, 6, 2)), bb.as_block(b))
w = a.copy(); w[1:4, :, 2:6:2] = b
eq('set_item strided', bb.to_numpy(y), w)
# BLAS-1 on lists with empty / 1-element / large blocks
blocks = [rng.standard_normal(s) for s in [(1,), (0, 3), (1000, 1001), (3, 3), (17,)]]
dev = [bb.as_block(b) for b in blocks]
n2 = bb.norm_many(dev); eq('norm_many', np.array(n2), np.array(np.sqrt(sum((b ** 2).sum() for b in blocks))), 1e-9)
ip = bb.inner_many(dev, dev); eq('inner_many', np.array(ip), np.array(sum((b ** 2).sum() for b in blocks)), 1e-6)
lc = bb.linear_combination_many(2.0, dev, -3.0, dev)
for got, b in zip(lc, blocks): eq('axpby', bb.to_numpy(got), -b, 1e-12)
eq('max_abs_many', np.array(bb.max_abs_many(dev)), np.array(max(np.abs(b).max() for b in blocks if b.size)))
# masks
m = rng.random(11) < 0.5
eq('apply_mask', bb.to_numpy(bb.apply_mask(x, m, 1)), a[:, m, :])
eq('apply_mask none', bb.to_numpy(bb.apply_mask(x, np.zeros(11, bool), 1)), a[:, np.zeros(11, bool), :])
small = rng.standard_normal((9, int(m.sum()), 7))
big = np.zeros((9, 11, 7)); big[:, m, :] = small
eq('enlarge_leg', bb.to_numpy(bb.enlarge_leg(bb.as_block(small), m, 1)), big)
# scale_axis on every axis incl. singleton
a4 = rng.standard_normal((4, 1, 6, 3)); x4 = bb.as_block(a4)
for ax in range(4):
    f = rng.standard_normal(a4.shape[ax])
    sh = [1] * 4; sh[ax] = -1
    eq(f'scale_axis {ax}', bb.to_numpy(bb.scale_axis(x4, bb.as_block(f), ax)), a4 * f.reshape(sh), 1e-15)
# combine / split legs round trip
a5 = rng.standard_normal((3, 4, 5, 2)); x5 = bb.as_block(a5)
c = bb.combine_legs(x5, [[1, 2]]); eq('combine_legs', bb.to_numpy(c), a5.reshape(3, 20, 2))
eq('split_legs', bb.to_numpy(bb.split_legs(c, [1], [[4, 5]])), a5)
eq('dagger', bb.to_numpy(bb.dagger(x5)), a5.transpose(3, 2, 1, 0))
print('blockops probe done; mismatches:', bad)
