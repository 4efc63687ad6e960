"""Does the second (LQ) preconditioning step pay for FULL-RANK blocks, as a function of their grading?  One 1024 x 1024 block and a
list of four 512 x 512 blocks, Gaussian times diag(logspace(0, -d, n)); run once plain and once with CYB_SVD_NOLQ=1.
    python scripts/lq_choice_probe.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)
kind = sys.argv[1] if len(sys.argv) > 1 else 'columns'
for d in (0, 2, 4, 6, 8, 10, 12, 14):
    for name, shapes in (('1024', [(1024, 1024)]), ('4x512', [(512, 512)] * 4), ('tall 1400x600', [(1400, 600)])):
        if kind == 'columns':
            mats = [rng.standard_normal(s) * np.logspace(0, -d, s[1]) for s in shapes]
        else:   # graded SPECTRUM with random singular vectors (a DMRG theta)
            mats = []
            for s in shapes:
                k = min(s)
                q1, _ = np.linalg.qr(rng.standard_normal((s[0], k)))
                q2, _ = np.linalg.qr(rng.standard_normal((s[1], k)))
                mats.append((q1 * np.logspace(0, -d, k)) @ q2.T)
        blocks = [bb.as_block(m) for m in mats]
        res, info = bb.matrix_svd_batched(blocks, return_info=True)
        bb.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); res = bb.matrix_svd_batched(blocks); bb.synchronize(); ts.append(time.perf_counter() - t0)
        err = 0.0
        for m, (U, S, Vh) in zip(mats, res):
            U, S, Vh = bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh)
            err = max(err, np.abs((U * S) @ Vh - m).max() / np.linalg.norm(m), np.abs(U.T @ U - np.eye(len(S))).max(), np.abs(S - np.linalg.svd(m, compute_uv=False)).max() / np.linalg.norm(m))
        sg = np.linalg.norm(np.linalg.qr(mats[0])[1], axis=1)
        print(f'[lq-probe] {kind} d={d:2d} {name}: {1e3 * min(ts):7.2f} ms sweeps {info} err {err:.1e}  row-norm ratio of R {sg.max() / sg.min():.1e}', flush=True)
