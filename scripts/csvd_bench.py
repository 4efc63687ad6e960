"""complex128 SVD / eigh / QR of large blocks: time and accuracy next to numpy on the host (SVD: the embedded route on the
float64 block engine beside the complex Jacobi kernels of csrc/csvd_large.hip)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from cyten_amd.block_backend import HipBlockBackend

bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(5)


def crandn(shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def timed(fn, reps=4):
    """best of `reps` single calls after a warm call (the host BLAS runs beside these: a mean would carry its clock ramps)"""
    fn(); bb.synchronize()
    best = 1e30
    for _ in range(reps):
        bb.synchronize()
        t0 = time.perf_counter()
        out = fn()
        bb.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


cases = [('full 256x256', crandn((256, 256))), ('full 512x512', crandn((512, 512))), ('full 1024x1024', crandn((1024, 1024))),
         ('theta-like 824x721 rank 412', crandn((824, 412)) @ crandn((412, 721))), ('tall 1442x360', crandn((1442, 360)))]
for name, a in cases:
    A = bb.as_block(a)
    t, (u, s, vh) = timed(lambda: bb.matrix_svd(A))
    t_old = timed(lambda: bb.matrix_svd_batched_complex_direct([A]))[0]   # the complex Jacobi kernels (round-1 route)
    u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
    t0 = time.perf_counter(); sr = np.linalg.svd(a, compute_uv=True, full_matrices=False)[1]; tc = time.perf_counter() - t0
    nrm = np.linalg.norm(a)
    k = min(a.shape)
    print(f'[csvd] {name}: {t*1e3:.1f} ms (complex kernels {t_old*1e3:.1f} ms; numpy {tc*1e3:.0f} ms -> {tc/t:.1f}x)  |dS| {np.abs(s-sr).max()/nrm:.1e}  recon {np.abs((u*s)@vh-a).max()/nrm:.1e}  '
          f'U {np.abs(u.conj().T@u-np.eye(k)).max():.1e}  V {np.abs(vh@vh.conj().T-np.eye(k)).max():.1e}', flush=True)
for n in (256, 512, 1024):
    z = crandn((n, n)); h = z + z.conj().T
    H = bb.as_block(h)
    t, (w, v) = timed(lambda: bb.eigh(H))
    t_old = timed(lambda: bb.eigh_batched([H], _embed=False))[0]
    w, v = bb.to_numpy(w), bb.to_numpy(v)
    t0 = time.perf_counter(); wr = np.linalg.eigh(h)[0]; tc = time.perf_counter() - t0
    nrm = np.linalg.norm(h)
    print(f'[ceigh] {n}: {t*1e3:.1f} ms (complex kernels {t_old*1e3:.1f} ms; numpy {tc*1e3:.0f} ms -> {tc/t:.1f}x)  |dw| {np.abs(w-wr).max()/nrm:.1e}  resid {np.abs(h@v-v*w).max()/nrm:.1e}  '
          f'V {np.abs(v.conj().T@v-np.eye(n)).max():.1e}', flush=True)
import scipy.linalg
for name, a in [('qr 512x512', crandn((512, 512))), ('qr 1442x360', crandn((1442, 360))), ('qr 1024x1024', crandn((1024, 1024)))]:
    A = bb.as_block(a)
    t, (q, r) = timed(lambda: bb.matrix_qr(A, False))
    q, r = bb.to_numpy(q), bb.to_numpy(r)
    t0 = time.perf_counter(); scipy.linalg.qr(a, mode='economic'); tc = time.perf_counter() - t0
    print(f'[cqr] {name}: {t*1e3:.1f} ms (scipy {tc*1e3:.0f} ms -> {tc/t:.1f}x)  recon {np.abs(q@r-a).max()/np.linalg.norm(a):.1e}  '
          f'Q {np.abs(q.conj().T@q-np.eye(q.shape[1])).max():.1e}', flush=True)
