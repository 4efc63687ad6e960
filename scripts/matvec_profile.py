"""cProfile of the host side of one H_eff matvec (development aid)."""
import sys, time, cProfile, pstats
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import krylov, workloads as wl
from helpers import to_device_tensor
bb = HipBlockBackend('cuda:0')
cfg = wl.config_heff(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 5, seed=11)
dev = {k: to_device_tensor(bb, v) for k, v in cfg.items()}
H = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'])
for _ in range(3):
    H.matvec(dev['theta'])
bb.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    H.matvec(dev['theta'])
t1 = time.perf_counter(); bb.synchronize(); t2 = time.perf_counter()
print(f'host issue time per matvec {1e2*(t1-t0):.2f} ms, wall {1e2*(t2-t0):.2f} ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    H.matvec(dev['theta'])
bb.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
