"""QR and SVD of extreme shapes (5000 x 64 ... 2049 x 2049, tall / wide / rank deficient) through the public batched entries: a sanity run of
the multi-workgroup panel kernel, the row-split strips and the per-block LQ rule beyond the sizes of the test-suite.  python scripts/extreme_shapes.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(3)
shapes = [(5000, 64), (64, 5000), (3000, 300), (300, 3000), (2500, 1200), (1537, 33), (1536, 1536), (4000, 32), (33, 4000), (2049, 2049)]
mats = [rng.standard_normal(s) for s in shapes]
mats[4] = rng.standard_normal((2500, 400)) @ rng.standard_normal((400, 1200))     # rank deficient, tall
for full in (False,):
    t0 = time.time()
    for a, (q, r) in zip(mats, bb.matrix_qr_batched([bb.as_block(a) for a in mats], full)):
        q, r = bb.to_numpy(q), bb.to_numpy(r)
        nrm = np.linalg.norm(a)
        e = [np.abs(q @ r - a).max() / nrm, np.abs(q.T @ q - np.eye(q.shape[1])).max(), np.abs(np.tril(r, -1)).max() / nrm]
        print(f'qr {a.shape}: recon {e[0]:.1e} Q {e[1]:.1e} tril {e[2]:.1e}', 'OK' if max(e) < 1e-10 else 'FAIL', flush=True)
    print('qr time', time.time() - t0)
t0 = time.time()
res, info = bb.matrix_svd_batched([bb.as_block(a) for a in mats], return_info=True)
for a, (u, s, vh), sw in zip(mats, res, info):
    u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
    nrm = np.linalg.norm(a); k = min(a.shape)
    e = [np.abs(s - np.linalg.svd(a, compute_uv=False)).max() / nrm, np.abs((u * s) @ vh - a).max() / nrm, np.abs(u.T @ u - np.eye(k)).max(), np.abs(vh @ vh.T - np.eye(k)).max()]
    print(f'svd {a.shape}: sweeps {sw} dS {e[0]:.1e} recon {e[1]:.1e} U {e[2]:.1e} V {e[3]:.1e}', 'OK' if max(e) < 1e-10 else 'FAIL', flush=True)
print('svd time', time.time() - t0)
