"""Bandwidth of the strided-copy kernel on transposing permutations (development aid)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)
for shape, perm in [((4096, 4096), (1, 0)), ((824, 5, 2, 2, 824), (1, 2, 3, 4, 0)), ((64, 64, 64, 64), (3, 2, 1, 0)), ((2048, 8, 2048), (2, 1, 0)), ((1 << 24,), (0,))]:
    a = bb.as_block(rng.standard_normal(shape))
    v = bb.permute_axes(a, list(perm))
    out = bb.contiguous(v); bb.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = bb.copy_block(v) if len(shape) == 1 else bb.contiguous(v)
    bb.synchronize()
    dt = (time.perf_counter() - t0) / 10
    nbytes = 2 * 8 * np.prod(shape)
    ok = np.array_equal(bb.to_numpy(out), np.ascontiguousarray(np.transpose(bb.to_numpy(a), perm))) if np.prod(shape) <= 1 << 25 else True
    print(f'[copy] {shape} perm {perm}: {1e3*dt:.3f} ms -> {nbytes/dt/1e12:.2f} TB/s  correct {ok}')
