"""cfg4 (SU(2) / SU(2)xU(1) FusionTreeBackend structure at chi=512, tests/golden/su2_chi512.npz): the coupled-sector GEMM list
of a compose (one grouped launch), the batched SVD of the same sector matrices and the F-move (`transform_blocks`, one launch per
tensor) on the device, with the per-block numpy / scipy loop of the reference's call pattern on the host beside them."""
import sys, time
import numpy as np
import scipy.linalg
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import su2_fixture as sf
from cyten_amd.block_backend import HipBlockBackend

bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)
z = sf.load()


def timed(fn, reps=5):
    fn(); bb.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    bb.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for which, lst in sf.compose_lists(z).items():
    a_np = [rng.standard_normal((m, k)) for m, k, n in lst]
    b_np = [rng.standard_normal((k, n)) for m, k, n in lst]
    a, b = [bb.as_block(x) for x in a_np], [bb.as_block(x) for x in b_np]
    flops = sum(2.0 * m * k * n for m, k, n in lst)
    t_dev = timed(lambda: bb.matrix_dot_grouped([[(x, y)] for x, y in zip(a, b)]))
    t0 = time.perf_counter(); [x @ y for x, y in zip(a_np, b_np)]; t_host = (time.perf_counter() - t0) * 1e3
    mats = [x @ y for x, y in zip(a_np, b_np)]
    blocks = [bb.as_block(m) for m in mats]
    t_svd = timed(lambda: bb.matrix_svd_batched(blocks), reps=3)
    t0 = time.perf_counter(); [scipy.linalg.svd(m, full_matrices=False) for m in mats]; t_hsvd = (time.perf_counter() - t0) * 1e3
    print(f'[su2] {which}: {len(lst)} coupled sectors (largest {max(lst)}), compose {flops / 1e9:.2f} GFLOP: device {t_dev:.3f} ms, host loop '
          f'{t_host:.1f} ms; batched SVD {t_svd:.2f} ms, scipy loop {t_hsvd:.1f} ms', flush=True)
    keys, rows, ncols, fwd, inv = sf.tree_move(z, which, ncols_of=lambda k: 64)
    shapes = [(r, c) for r, c in zip(rows, ncols)]
    blks = [bb.as_block(rng.standard_normal(sh)) for sh in shapes]
    t_f = timed(lambda: bb.transform_blocks(blks, shapes, fwd))
    print(f'[su2] {which}: F-move over {len(blks)} blocks / {len(fwd)} tree blocks / {sum(len(u[-1]) for u in fwd)} terms: device {t_f:.3f} ms', flush=True)
