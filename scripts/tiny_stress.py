"""Many-tiny-blocks stress (cfg4-like: FusionTreeBackend has one small 2-D block per coupled sector)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)
n = 400
shapes = [(int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 40))) for _ in range(n)]
A = [bb.as_block(rng.standard_normal((m, k))) for m, k, _ in shapes]
B = [bb.as_block(rng.standard_normal((k, nn))) for _, k, nn in shapes]
groups = [[(a, b)] for a, b in zip(A, B)]
def timeit(fn, reps=5):
    fn(); bb.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); bb.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), r
t, outs = timeit(lambda: bb.matrix_dot_grouped(groups))
print(f'[tiny] {n} GEMMs <=40^3 grouped: {1e3*t:.2f} ms ({n/t/1e3:.0f} kGEMM/s)')
t1, _ = timeit(lambda: [bb.matrix_dot(a, b) for a, b in zip(A[:50], B[:50])], reps=2)
print(f'[tiny] 50 GEMMs one call each (the reference call pattern): {1e3*t1:.2f} ms -> {1e3*t1/50:.3f} ms per GEMM')
M = [bb.as_block(rng.standard_normal((m, nn))) for m, _, nn in shapes]
t, res = timeit(lambda: bb.matrix_svd_batched(M), reps=3)
print(f'[tiny] {n} SVDs <=40x40 batched: {1e3*t:.2f} ms')
import scipy.linalg
hm = [bb.to_numpy(x) for x in M]
t0 = time.perf_counter(); [scipy.linalg.svd(x, full_matrices=False) for x in hm]; t1c = time.perf_counter() - t0
print(f'[tiny] cpu scipy loop: {1e3*t1c:.2f} ms')
worst = 0
for x, (U, S, Vh) in zip(hm[:40], res[:40]):
    U, S, Vh = bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh)
    worst = max(worst, np.abs((U * S) @ Vh - x).max())
print('[tiny] worst recon', worst)
t, res = timeit(lambda: bb.matrix_qr_batched(M), reps=3)
print(f'[tiny] {n} QRs batched: {1e3*t:.2f} ms')
H = [bb.as_block((lambda a: a + a.T)(rng.standard_normal((m, m)))) for m, _, _ in shapes]
t, res = timeit(lambda: bb.eigh_batched(H), reps=3)
print(f'[tiny] {n} eighs batched: {1e3*t:.2f} ms')
