"""cfg5 (CTMRG D=6 chi=256): per-sector eigh of the enlarged corner + QR of the projector blocks."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import workloads as wl
import scipy.linalg
bb = HipBlockBackend('cuda:0')
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
herm, tall = wl.config_ctmrg_blocks(scale=scale)
print('eigh blocks', [h.shape[0] for h in herm]); print('qr blocks', [t.shape for t in tall])
H = [bb.as_block(h) for h in herm]; T = [bb.as_block(t) for t in tall]
for name, fn in (('eigh', lambda: bb.eigh_batched(H)), ('qr', lambda: bb.matrix_qr_batched(T))):
    fn(); bb.synchronize()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); res = fn(); bb.synchronize(); ts.append(time.perf_counter() - t0)
    print(f'[cfg5] {name}: {1e3*min(ts):.1f} ms')
t0 = time.perf_counter(); [np.linalg.eigh(h) for h in herm]; t1 = time.perf_counter()
[scipy.linalg.qr(t, mode='economic') for t in tall]; t2 = time.perf_counter()
print(f'[cfg5] cpu eigh {1e3*(t1-t0):.1f} ms, cpu qr {1e3*(t2-t1):.1f} ms')
W, V = res if False else (None, None)
w, v = bb.eigh_batched(H[:1])[0]
w, v = bb.to_numpy(w), bb.to_numpy(v)
print('eigh err', np.abs(w - np.linalg.eigvalsh(herm[0])).max() / np.abs(w).max(), np.abs(herm[0] @ v - v * w).max() / np.abs(w).max())
