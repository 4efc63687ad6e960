"""Cost of lq_check_kernel on large graded blocks (many legitimate rows below the rank threshold): run under rocprofv3 --stats."""
import sys; sys.path.insert(0, '.')
import numpy as np, time
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)
for dec in (8, 14, 18):
    a = rng.standard_normal((1442, 1236)) * np.logspace(0, -dec, 1236)
    q, _ = np.linalg.qr(rng.standard_normal((1236, 1236)))
    a = a @ q
    blk = bb.as_block(a)
    bb.matrix_svd_batched([blk])
    bb.ctx.synchronize()
    t0 = time.perf_counter()
    res, info = bb.matrix_svd_batched([blk], return_info=True)
    bb.ctx.synchronize()
    t = time.perf_counter() - t0
    u, s, vh = (bb.to_numpy(x) for x in res[0])
    print(f'graded over {dec} decades: {t * 1e3:.1f} ms, sweeps {info}, recon {np.abs((u * s) @ vh - a).max():.1e}, '
          f'U {np.abs(u.T @ u - np.eye(1236)).max():.1e}, V {np.abs(vh @ vh.T - np.eye(1236)).max():.1e}', flush=True)
