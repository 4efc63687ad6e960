import sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)
mats = [bb.as_block(rng.standard_normal((int(rng.integers(2, 41)), int(rng.integers(2, 41))))) for _ in range(400)]
for _ in range(3):
    bb.matrix_svd_batched(mats)
bb.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    bb.matrix_svd_batched(mats)
bb.synchronize()
print('per call %.2f ms' % (1e2 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    bb.matrix_svd_batched(mats)
bb.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(10)
