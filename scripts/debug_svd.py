import sys
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(12345)

def diag(name, m):
    (U, S, Vh), info = (lambda r: (r[0][0], r[1]))(bb.matrix_svd_batched([bb.as_block(m)], return_info=True))
    U, S, Vh = bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh)
    k = len(S)
    eu = np.abs(U.T @ U - np.eye(k)); ev = np.abs(Vh @ Vh.T - np.eye(k))
    iu = np.unravel_index(np.argmax(eu), eu.shape); iv = np.unravel_index(np.argmax(ev), ev.shape)
    sref = np.linalg.svd(m, compute_uv=False)
    thresh = S[0] * max(m.shape) * 2.2e-16
    print(f'{name}: shape {m.shape} sweeps {info} UtU err {eu.max():.2e} at {iu} (S={S[iu[0]]:.2e},{S[iu[1]]:.2e}) '
          f'VVt err {ev.max():.2e} at {iv} (S={S[iv[0]]:.2e},{S[iv[1]]:.2e}) dS {np.abs(S-sref).max()/sref[0]:.1e} '
          f'recon {np.abs((U*S)@Vh-m).max()/sref[0]:.1e} n_below_thresh {(S<=thresh).sum()} thresh {thresh:.1e}')
    bad = np.argwhere(eu > 1e-10)
    if len(bad):
        cols = sorted(set(bad[:, 0].tolist()))
        print('   bad U columns:', cols[:20], 'S there:', S[cols[:20]])

graded = rng.standard_normal((70, 70)) * np.logspace(0, -14, 70)[None, :]
diag('graded cols', graded)
diag('graded rows', graded.T.copy())
q1 = np.linalg.qr(rng.standard_normal((90, 90)))[0]; q2 = np.linalg.qr(rng.standard_normal((90, 90)))[0]
diag('exp decay', (q1 * np.exp(-np.arange(90.0))) @ q2)
diag('lowrank 80x60 r3', rng.standard_normal((80, 3)) @ rng.standard_normal((3, 60)))
diag('theta-like 120x100 r50', rng.standard_normal((120, 50)) @ rng.standard_normal((50, 100)))
diag('theta-like 300x280 r150', rng.standard_normal((300, 150)) @ rng.standard_normal((150, 280)))
diag('rank1', np.outer(rng.standard_normal(50), rng.standard_normal(77)))
