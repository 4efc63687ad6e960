"""Batched SVD of lists that oversubscribe the persistent sweep kernel (more block pairs than CUs: workgroups own several
entries) or mix one large block with many small ones: LAPACK parity of every block, time per call, sweeps."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import check_svd_invariants
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(3)
def run(name, mats):
    blocks = [bb.as_block(m) for m in mats]
    bb.matrix_svd_batched(blocks)
    bb.synchronize()
    t0 = time.perf_counter()
    res, info = bb.matrix_svd_batched(blocks, return_info=True)
    bb.synchronize()
    dt = time.perf_counter() - t0
    for m, (U, S, Vh) in zip(mats, res):
        check_svd_invariants(m, bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh), 1e-10, sref=np.linalg.svd(m, compute_uv=False))
    print(f'{name}: {len(mats)} blocks, {dt * 1e3:.1f} ms, sweeps {min(info)}..{max(info)} OK', flush=True)
run('120 x 130^2', [rng.standard_normal((130, 130)) for _ in range(120)])
run('1 x 1442^2 rank-deficient + 150 x 100^2', [rng.standard_normal((1442, 700)) @ rng.standard_normal((700, 1442))] + [rng.standard_normal((100, 100)) for _ in range(150)])
run('40 x 300x200 + 40 x 200x300', [rng.standard_normal((300, 200)) for _ in range(40)] + [rng.standard_normal((200, 300)) for _ in range(40)])
run('300 x 70x90 graded', [rng.standard_normal((70, 90)) * np.logspace(0, -12, 90) for _ in range(300)])
