"""Whole-loop timing of the toy two-site DMRG (tests/toy_dmrg.py: Heisenberg chain, U(1)) on the HIP backend and on the
numpy stand-in backend (the reference's per-block call pattern on the host): seconds per sweep at growing bond dimension."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import toy_dmrg as td
from numpy_backend import NumpyGroupedBackend
from cyten_amd.block_backend import HipBlockBackend

L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
chis = [int(x) for x in sys.argv[2:]] or [64, 256]
bb, nb = HipBlockBackend('cuda:0'), NumpyGroupedBackend()
for chi in chis:
    res = {}
    for name, backend in (('hip', bb), ('numpy', nb)):
        td.dmrg(backend, td.heisenberg_model(L, 1.0), chi_max=chi, svd_min=1e-30, n_sweeps=1, lanczos_options=dict(N_max=4))   # compile / warm caches
        t0 = time.perf_counter()
        n_sw = 11 if chi <= 256 else 13   # the bond dimension doubles per sweep at most: time the LAST two of eleven sweeps
        E, psi, ts = td.dmrg(backend, td.heisenberg_model(L, 1.0), chi_max=chi, svd_min=1e-30, n_sweeps=n_sw, lanczos_options=dict(N_max=6), sweep_times=True)
        if name == 'hip':
            bb.synchronize()
        res[name] = (sum(ts[-2:]) / 2, E, max(t.legs[2].dim for t in psi))
    (tg, Eg, dg), (tc, Ec, dc) = res['hip'], res['numpy']
    print(f'[dmrg] L={L} chi_max={chi} (reached {dg}): hip {tg:.2f} s/sweep, numpy stand-in {tc:.2f} s/sweep -> {tc/tg:.1f}x; '
          f'E/L hip {Eg/L:.10f} numpy {Ec/L:.10f}', flush=True)
