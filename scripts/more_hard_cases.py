"""Hard inputs for eigh / QR / GEMM (probe; regressions go to tests/)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(11)
def eigh_check(name, h):
    try:
        w, v = [bb.to_numpy(x) for x in bb.eigh(bb.as_block(h))]
        wr = np.linalg.eigvalsh(h)
        sc = max(np.abs(h).max(), 1e-300)
        print(f'eigh {name:28s} n={h.shape[0]:4d}: dW {np.abs(w - wr).max() / sc:.1e} resid {np.abs(h @ v - v * w).max() / sc:.1e} ortho {np.abs(v.T @ v - np.eye(len(w))).max():.1e}')
    except Exception as e:
        print(f'eigh {name:28s}: EXCEPTION {type(e).__name__} {str(e)[:70]}')
for n in (6, 40, 150):
    g = rng.standard_normal((n, n)); h = g + g.T
    eigh_check('random', h)
    eigh_check('identity', np.eye(n))
    eigh_check('zero', np.zeros((n, n)))
    eigh_check('all ones', np.ones((n, n)))
    eigh_check('negative definite', -(g @ g.T) - np.eye(n))
    q, _ = np.linalg.qr(g)
    eigh_check('degenerate (3 values)', (q * np.repeat([-1.0, 0.5, 2.0], [n // 3, n // 3, n - 2 * (n // 3)])) @ q.T)
    eigh_check('graded 1..1e-12', (q * np.logspace(0, -12, n)) @ q.T)
    eigh_check('diag + tiny offdiag', np.diag(np.arange(1.0, n + 1)) + 1e-12 * h)
def qr_check(name, a):
    try:
        Q, R = [bb.to_numpy(x) for x in bb.matrix_qr(bb.as_block(a), False)]
        sc = max(np.abs(a).max(), 1e-300)
        print(f'qr   {name:28s} {a.shape}: recon {np.abs(Q @ R - a).max() / sc:.1e} ortho {np.abs(Q.T @ Q - np.eye(Q.shape[1])).max():.1e}')
    except Exception as e:
        print(f'qr   {name:28s}: EXCEPTION {type(e).__name__} {str(e)[:70]}')
for shp in ((30, 30), (200, 120), (120, 200)):
    qr_check('zero', np.zeros(shp))
    qr_check('identity-like', np.eye(*shp))
    z = rng.standard_normal(shp); z[:, ::3] = 0
    qr_check('zero columns', z)
    z = rng.standard_normal(shp); z[::2, :] = 0
    qr_check('zero rows', z)
# GEMM specials
a = rng.standard_normal((70, 50)); b = rng.standard_normal((50, 90))
a2 = a.copy(); a2[3, 4] = np.nan
c = bb.to_numpy(bb.matrix_dot(bb.as_block(a2), bb.as_block(b)))
print('gemm NaN propagates only into row 3:', bool(np.isnan(c[3]).all() and not np.isnan(np.delete(c, 3, axis=0)).any()))
a3 = a.copy(); a3[5, 6] = np.inf
c = bb.to_numpy(bb.matrix_dot(bb.as_block(a3), bb.as_block(b)))
print('gemm inf row 5 non-finite, others finite:', bool(not np.isfinite(c[5]).all() and np.isfinite(np.delete(c, 5, axis=0)).all()))
c = bb.to_numpy(bb.matrix_dot(bb.as_block(1e200 * a), bb.as_block(1e-200 * b)))
print('gemm scaled operands err', np.abs(c - a @ b).max())
