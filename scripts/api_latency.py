"""Host-side latency of single-block operator calls (the reference's one-block-at-a-time API, numpy.cpp) on small blocks:
microseconds per call, device backend beside numpy.  Finds operations whose host path is out of proportion."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd.deferred import DeferredBlockBackend

bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)
a_np, b_np = rng.standard_normal((64, 64)), rng.standard_normal((64, 64))
t_np = rng.standard_normal((8, 6, 10, 4))
a, b, t4 = bb.as_block(a_np), bb.as_block(b_np), bb.as_block(t_np)
mask = np.arange(64) % 3 != 0


def us(fn, reps=200, sync=True):
    for _ in range(5):
        fn()
    bb.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync:
        bb.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


rows = [
    ('matrix_dot 64^3', lambda: bb.matrix_dot(a, b), lambda: a_np @ b_np),
    ('tdot [8,6,10,4] x itself over 2 axes', lambda: bb.tdot(t4, t4, [2, 3], [2, 3]), lambda: np.tensordot(t_np, t_np, ([2, 3], [2, 3]))),
    ('permute_axes + reshape (views)', lambda: bb.reshape(bb.permute_axes(t4, [2, 0, 3, 1]), (10, 8, 24)) if False else bb.permute_axes(t4, [2, 0, 3, 1]), lambda: t_np.transpose(2, 0, 3, 1)),
    ('copy of a permuted view', lambda: bb.contiguous(bb.permute_axes(t4, [2, 0, 3, 1])), lambda: np.ascontiguousarray(t_np.transpose(2, 0, 3, 1))),
    ('linear_combination', lambda: bb.linear_combination(0.5, a, 2.0, b), lambda: 0.5 * a_np + 2.0 * b_np),
    ('mul', lambda: bb.mul(3.0, a), lambda: 3.0 * a_np),
    ('norm (host scalar)', lambda: bb.norm(a), lambda: np.linalg.norm(a_np)),
    ('inner (host scalar)', lambda: bb.inner(a, b, True), lambda: np.vdot(a_np, b_np)),
    ('apply_mask axis 1', lambda: bb.apply_mask(a, mask, 1), lambda: a_np[:, mask]),
    ('scale_axis', lambda: bb.scale_axis(a, bb.as_block(a_np[0]), 1) if False else bb.scale_axis(a, a_row, 1), lambda: a_np * a_np[0][None, :]),
    ('get_item slice (view)', lambda: bb.get_item(a, (slice(3, 40), slice(None, None, 2))), lambda: a_np[3:40, ::2]),
    ('matrix_svd 64x64', lambda: bb.matrix_svd(a), lambda: np.linalg.svd(a_np, full_matrices=False)),
    ('matrix_qr 64x64', lambda: bb.matrix_qr(a, False), lambda: np.linalg.qr(a_np)),
    ('eigh 64x64', lambda: bb.eigh(bb.as_block(a_np + a_np.T)) if False else bb.eigh(a_sym), lambda: np.linalg.eigh(a_np + a_np.T)),
]
a_row = bb.as_block(a_np[0].copy())
a_sym = bb.as_block(a_np + a_np.T)
print(f'{"operation":44s} {"device us":>10s} {"numpy us":>10s}')
for name, fd, fh in rows:
    reps = 30 if 'svd' in name or 'qr' in name or 'eigh' in name else 200
    td = us(fd, reps)
    t0 = time.perf_counter()
    for _ in range(reps):
        fh()
    th = (time.perf_counter() - t0) / reps * 1e6
    print(f'{name:44s} {td:10.1f} {th:10.1f}', flush=True)
