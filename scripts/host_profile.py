"""cProfile of the host side of combine_legs / truncation gather / split_legs / compose (development aid)."""
import sys, cProfile, pstats
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import abelian as ab, workloads as wl
bb = HipBlockBackend('cuda:0')
A, B = wl.config_u1_mps(4096)
a = ab.AbelianTensor.from_spec(bb, A); b = ab.AbelianTensor.from_spec(bb, B)
theta = ab.compose(bb, a, b, 1)
mv = ab.combine_legs_to_matrix(bb, theta, 2)
usv = bb.matrix_svd_batched(mv.blocks)
S = [x[1] for x in usv]
masks, err, nn = ab.truncate_singular_values(bb, S, chi_max=4096)
items = [(x[0], m, 1) for x, m in zip(usv, masks)] + [(s, m, 0) for s, m in zip(S, masks)] + [(x[2], m, 0) for x, m in zip(usv, masks)]
kept = bb.mask_gather_many(items)
bb.synchronize()
def work():
    for _ in range(20):
        t = ab.compose(bb, a, b, 1)
        m = ab.combine_legs_to_matrix(bb, t, 2)
        k = bb.mask_gather_many(items)
        ab.split_matrix_legs(bb, m, k[2 * len(usv):], 'cols')
    bb.synchronize()
pr = cProfile.Profile(); pr.enable(); work(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
