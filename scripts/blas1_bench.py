"""Kernel-time check of the BLAS-1 / elementwise block-list kernels on large blocks (run under rocprofv3)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
n = 1 << 24
x = bb.random_normal((n,), seed=1); y = bb.random_normal((n,), seed=2)
for _ in range(3):
    bb.inner_many([x], [y]); bb.norm_many([x]); bb.linear_combination_many(2.0, [x], 3.0, [y]); bb.mul_many(0.5, [x]); bb.max_abs_many([x])
a = bb.random_normal((4096, 4096), seed=3); f = bb.random_normal((4096,), seed=4)
for _ in range(3):
    bb.scale_axis(a, f, 1); bb.scale_axis(a, f, 0)
m = np.random.default_rng(0).random(4096) < 0.5
for _ in range(3):
    bb.apply_mask(a, m, 1); bb.apply_mask(a, m, 0)
bb.synchronize(); print('done')
