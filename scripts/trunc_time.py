import sys, time
sys.path.insert(0, '.')
import numpy as np
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)
for n in (8192, 16384, 32768, 65536):
    S = [bb.as_block(np.sort(np.abs(rng.standard_normal(n // 32)))[::-1].copy()) for _ in range(32)]
    bb.truncate_select(S, chi_max=4096); bb.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        bb.truncate_select(S, chi_max=4096)
    bb.synchronize()
    print(f'[trunc] n={n}: {(time.perf_counter()-t0)/5*1e3:.2f} ms per call', flush=True)
