"""eigh of the matrices in an .npz (arr_0, arr_1, ...) in ONE batched call, errors per matrix (|dw| / resid / orthogonality, relative as in
scripts/svd_fuzz.py).  `CYB_JACOBI_TRACE=1 python scripts/eigh_one.py file.npz` prints the off-norm history of the iteration."""
import sys

import numpy as np

sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend

bb = HipBlockBackend('cuda:0')
d = np.load(sys.argv[1])
hs = [d[k] for k in d.files]
alone = '--alone' in sys.argv
lists = [[h] for h in hs] if alone else [hs]
for lst in lists:
    for h, (w, v) in zip(lst, bb.eigh_batched([bb.as_block(h) for h in lst])):
        w, v = bb.to_numpy(w), bb.to_numpy(v)
        nrm = max(np.abs(h).max() * h.shape[0], 1e-300)
        g = np.abs(v.conj().T @ v - np.eye(h.shape[0]))
        i, j = np.unravel_index(np.argmax(g), g.shape)
        print(f'[eigh-one] n {h.shape[0]}: dw {np.abs(w - np.linalg.eigvalsh(h)).max() / nrm:.1e} resid {np.abs(h @ v - v * w).max() / nrm:.1e} '
              f'V {g.max():.1e} at ({i}, {j}) w = {w[i]:.3e}, {w[j]:.3e}', flush=True)
