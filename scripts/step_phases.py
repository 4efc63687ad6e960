"""Phase breakdown of one bench step (synchronised between phases; development aid)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import abelian as ab, workloads as wl
bb = HipBlockBackend('cuda:0')
A, B = wl.config_u1_mps(4096)
st = bench.ThetaStep(bb, A, B, 4096)
for _ in range(2):
    st.step(); bb.synchronize()
a, b = st.a, st.b if hasattr(st, 'a') else (None, None)
def T(f):
    bb.synchronize(); t0 = time.perf_counter(); r = f(); bb.synchronize(); return r, 1e3 * (time.perf_counter() - t0)
theta, t1 = T(lambda: ab.compose(bb, st.a, st.b, 1))
mv, t2 = T(lambda: ab.combine_legs_to_matrix(bb, theta, 2))
usv, t3 = T(lambda: bb.matrix_svd_batched(mv.blocks))
S = [x[1] for x in usv]
(masks, err, nn), t4 = T(lambda: ab.truncate_singular_values(bb, S, chi_max=4096))
kept, t5 = T(lambda: bb.mask_gather_many([(x[0], m, 1) for x, m in zip(usv, masks)] + [(s, m, 0) for s, m in zip(S, masks)] + [(x[2], m, 0) for x, m in zip(usv, masks)]))
_, t6 = T(lambda: st.step())
print(f'[phases] compose {t1:.2f} ms, combine_legs {t2:.2f}, svd {t3:.2f}, truncate {t4:.2f}, gather {t5:.2f}; whole step {t6:.2f}')
