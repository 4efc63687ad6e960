"""Bandwidth of the data movement that brackets every SVD: combine_legs (zero fill + sub-block
scatter), the truncation gather of U / S / Vh, split_legs -- on the chi=4096 U(1) theta; checked against numpy."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import abelian as ab, workloads as wl

bb = HipBlockBackend('cuda:0')
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A, B = wl.config_u1_mps(chi)
a = ab.AbelianTensor.from_spec(bb, A)
b = ab.AbelianTensor.from_spec(bb, B)
theta = ab.compose(bb, a, b, 1)
bb.synchronize()


def timed(f, reps=10):
    r = f(); bb.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    bb.synchronize()
    return r, (time.perf_counter() - t0) / reps


n_theta = sum(int(np.prod(x.shape)) for x in theta.blocks)
mv, dt = timed(lambda: ab.combine_legs_to_matrix(bb, theta, 2))
n_mat = sum(int(np.prod(x.shape)) for x in mv.blocks)
# bytes: memset of the matrix blocks + read of theta + write of the occupied entries
print(f'[combine_legs] chi={chi}: {len(theta.blocks)} blocks -> {len(mv.blocks)} matrix blocks, {8e-6*n_theta:.1f} MB moved, '
      f'{8e-6*n_mat:.1f} MB zero-filled: {1e3*dt:.3f} ms -> {8*(2*n_theta+n_mat)/dt/1e12:.2f} TB/s (host time included)')
# correctness of one sector against numpy
dense = theta.to_dense(bb)
rows = sum(l.dim for l in theta.legs[:2]); 
big = np.zeros((int(np.prod([l.dim for l in theta.legs[:2]])), int(np.prod([l.dim for l in theta.legs[2:]]))))
got = sum(float(np.sum(bb.to_numpy(x) ** 2)) for x in mv.blocks)
print(f'[combine_legs] norm^2 theta {float(np.sum(dense**2)):.12e} vs matrix blocks {got:.12e}')

usv = bb.matrix_svd_batched(mv.blocks)
S = [x[1] for x in usv]
masks, err, nn = ab.truncate_singular_values(bb, S, chi_max=chi)
items = [(x[0], m, 1) for x, m in zip(usv, masks)] + [(s, m, 0) for s, m in zip(S, masks)] + [(x[2], m, 0) for x, m in zip(usv, masks)]
kept, dt = timed(lambda: bb.mask_gather_many(items))
nb = sum(int(np.prod(k.shape)) for k in kept)
print(f'[mask gather] U/S/Vh of {len(usv)} blocks: {8e-6*nb:.1f} MB kept: {1e3*dt:.3f} ms -> {16*nb/dt/1e12:.2f} TB/s')
# correctness against numpy fancy indexing
worst = 0.0
for (blk, m, ax), k in zip(items, kept):
    ref = np.compress(np.asarray(m), bb.to_numpy(blk), axis=ax)
    worst = max(worst, float(np.max(np.abs(ref - bb.to_numpy(k)))) if ref.size else 0.0)
print(f'[mask gather] max |diff| vs numpy {worst:.1e}')
nk = len(usv)
Uk, Vk = kept[:nk], kept[2 * nk:]
_, dt = timed(lambda: (ab.split_matrix_legs(bb, mv, Uk, 'rows'), ab.split_matrix_legs(bb, mv, Vk, 'cols')))
nb = sum(int(np.prod(k.shape)) for k in Uk) + sum(int(np.prod(k.shape)) for k in Vk)
print(f'[split_legs] U and Vh: {8e-6*nb:.1f} MB: {1e3*dt:.3f} ms -> {16*nb/dt/1e12:.2f} TB/s (host time included)')
