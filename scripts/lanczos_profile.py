"""cProfile + phase timing of the device Lanczos iteration (development aid)."""
import sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import abelian as ab, krylov, workloads as wl
from helpers import to_device_tensor
bb = HipBlockBackend('cuda:0')
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = wl.config_heff(chi, 5, seed=11)
dev = {k: to_device_tensor(bb, v) for k, v in cfg.items()}
H = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'])
krylov.lanczos(bb, H, dev['theta'], dict(N_max=3)); bb.synchronize()
th = dev['theta']
def T(f, reps=5):
    f(); bb.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    bb.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps
print('blocks', len(th.blocks), 'elements', sum(b.size for b in th.blocks))
print('matvec %.2f ms' % T(lambda: H.matvec(th)))
print('inner %.3f ms' % T(lambda: ab.inner(bb, th, th)))
print('norm %.3f ms' % T(lambda: ab.norm(bb, th)))
print('lincomb %.3f ms' % T(lambda: ab.linear_combination(bb, 1.0, th, -0.5, th)))
print('scale %.3f ms' % T(lambda: ab.scale(bb, 0.5, th)))
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); E0, psi, N = krylov.lanczos(bb, H, th, dict(N_max=10)); bb.synchronize(); t = time.perf_counter() - t0
pr.disable()
print('lanczos N=%d: %.1f ms' % (N, 1e3 * t))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
