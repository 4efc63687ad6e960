"""Hard inputs for the batched SVD / eigh (scaling, clusters, graded spectra, special shapes)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(5)
def check(name, a):
    try:
        U, S, Vh = bb.matrix_svd(bb.as_block(a))
        U, S, Vh = bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh)
        sref = np.linalg.svd(a, compute_uv=False)
        nrm = max(np.abs(a).max(), 1e-300)
        rec = np.abs((U * S) @ Vh - a).max() / nrm
        ds = np.abs(S - sref).max() / max(sref[0], 1e-300)
        ortho = max(np.abs(U.T @ U - np.eye(U.shape[1])).max(), np.abs(Vh @ Vh.T - np.eye(Vh.shape[0])).max())
        print(f'{name:38s} shape {a.shape}: recon {rec:.1e} dS {ds:.1e} ortho {ortho:.1e} finite {np.isfinite(S).all()}')
    except Exception as e:
        print(f'{name:38s} shape {a.shape}: EXCEPTION {type(e).__name__}: {str(e)[:80]}')
for n in (40, 200):
    g = rng.standard_normal((n, n))
    check('scaled 1e+150', 1e150 * g)
    check('scaled 1e-150', 1e-150 * g)
    check('scaled 1e+200', 1e200 * g)
    check('identity', np.eye(n))
    check('all ones (rank 1)', np.ones((n, n)))
    q1, _ = np.linalg.qr(rng.standard_normal((n, n))); q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
    check('clustered (half 1.0, half 1e-3)', (q1 * np.r_[np.ones(n // 2), 1e-3 * np.ones(n - n // 2)]) @ q2)
    check('graded 1e0..1e-15', (q1 * np.logspace(0, -15, n)) @ q2)
    check('graded columns', g * np.logspace(0, -12, n)[None, :])
    check('tall 4n x n', rng.standard_normal((4 * n, n)))
    check('wide n x 4n', rng.standard_normal((n, 4 * n)))
    z = g.copy(); z[:, n // 3] = 0; z[n // 2, :] = 0
    check('zero row and column', z)
check('1 x 1', np.array([[3.0]]))
check('1 x 7', rng.standard_normal((1, 7)))
check('7 x 1', rng.standard_normal((7, 1)))
check('with NaN', np.array([[1.0, np.nan], [0.0, 1.0]]))
