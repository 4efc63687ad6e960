"""One batched SVD of the rank-deficient 1442 x 1442 theta-like block, a few repetitions (for rocprofv3 --kernel-trace)."""
import sys

import numpy as np

sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend

bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1442
a = bb.as_block(rng.standard_normal((n, n // 2)) @ rng.standard_normal((n // 2, n)))
for _ in range(4):
    bb.matrix_svd_batched([a])
bb.synchronize()
print('done')
