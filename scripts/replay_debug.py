import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import krylov, workloads as wl, abelian as ab
from helpers import to_device_tensor
bb = HipBlockBackend('cuda:0')
cfg = wl.config_heff(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 5, seed=11)
dev = {k: to_device_tensor(bb, v) for k, v in cfg.items()}
H = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'])
th = dev['theta']
for i in range(4):
    t0 = time.perf_counter(); out = H.matvec(th); t1 = time.perf_counter(); bb.synchronize(); t2 = time.perf_counter()
    print(i, 'issue %.2f ms wall %.2f ms' % (1e3*(t1-t0), 1e3*(t2-t0)), 'replayed', H.n_replayed, [(r.valid, r.reason, len(r.plan)) for r in H._recordings.values()])
v = ab.scale(bb, 0.5, th)
for i in range(3):
    t0 = time.perf_counter(); out = H.matvec(v); t1 = time.perf_counter(); bb.synchronize(); t2 = time.perf_counter()
    print('pooled', i, 'issue %.2f ms wall %.2f ms' % (1e3*(t1-t0), 1e3*(t2-t0)), 'replayed', H.n_replayed, len(H._recordings))
    v = ab.scale(bb, 0.5, out) if False else ab.scale(bb, 0.5, v)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    H.matvec(th)
bb.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
# time the C calls of one replay
import ctypes as C, numpy as np
from cyten_amd import _lib
rec = [r for r in H._recordings.values()][0]
lib, handle = bb.lib, bb.ctx.handle
bb.synchronize()
for ev in rec.plan:
    if ev[0] == 'call':
        _, name, arrays, scalars, fields = ev
        t0 = time.perf_counter()
        if name == 'cyb_copy_strided_batched':
            lib.cyb_copy_strided_batched(handle, arrays[0].ctypes.data_as(C.POINTER(_lib.CopyDesc)), scalars[0], scalars[1])
        else:
            lib.cyb_gemm_grouped_enqueue_f64(handle, arrays[0].ctypes.data_as(C.POINTER(_lib.GemmProb)), scalars[0],
                                             arrays[1].ctypes.data_as(C.POINTER(_lib.GemmSeg)), scalars[1])
        t1 = time.perf_counter(); bb.synchronize()
        print(name, scalars, 'host %.3f ms' % (1e3 * (t1 - t0)))
