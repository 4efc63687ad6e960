"""Lanczos matvec group on the device: H_eff matvec = 4 grouped-GEMM composes + 4 leg
permutations; timing against the numpy per-block path (the oracle-based stand-in backend) on the host."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import abelian as ab, krylov, workloads as wl
from helpers import to_device_tensor
from numpy_backend import NumpyGroupedBackend

bb = HipBlockBackend('cuda:0')
nbk = NumpyGroupedBackend()
chis = [int(x) for x in sys.argv[1:]] or [1024, 4096]
for chi in chis:
    cfg = wl.config_heff(chi, 5, seed=11)
    dev = {k: to_device_tensor(bb, v) for k, v in cfg.items()}
    H = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'])
    out = H.matvec(dev['theta']); bb.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = H.matvec(dev['theta']); bb.synchronize(); ts.append(time.perf_counter() - t0)
    t_gpu = min(ts)
    fl = H.flops_per_matvec
    # host reference path, one matvec
    cpu = {k: to_device_tensor(nbk, v) for k, v in cfg.items()}
    Hc = krylov.HEffective(nbk, cpu['LP'], cpu['W1'], cpu['W2'], cpu['RP'])
    Hc.matvec(cpu['theta'])
    t0 = time.perf_counter(); ref = Hc.matvec(cpu['theta']); t_cpu = time.perf_counter() - t0
    err = max(np.abs(bb.to_numpy(x) - y).max() for x, y in zip(out.blocks, ref.blocks)) / max(np.abs(y).max() for y in ref.blocks)
    print(f'[heff] chi={chi} D=5: {fl/1e9:.1f} GFLOP per matvec; device {1e3*t_gpu:.2f} ms ({fl/t_gpu/1e12:.2f} TFLOP/s), '
          f'host numpy path {1e3*t_cpu:.0f} ms ({fl/t_cpu/1e9:.0f} GFLOP/s) -> {t_cpu/t_gpu:.0f}x; rel err {err:.1e}', flush=True)
    t0 = time.perf_counter()
    E0, psi, N = krylov.lanczos(bb, H, dev['theta'], dict(N_max=10))
    bb.synchronize(); t_l = time.perf_counter() - t0
    print(f'[lanczos] chi={chi}: N={N}, E0={E0:.6e}, {1e3*t_l:.1f} ms total ({1e3*t_l/N:.2f} ms per iteration)', flush=True)
