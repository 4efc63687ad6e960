"""Host profile of the toy two-site DMRG (tests/toy_dmrg.py) at a fixed bond dimension: cProfile of the LAST sweeps
only (bond dimension saturated, every launch sequence replayed), seconds per sweep and per bond update.
    python3 scripts/dmrg_profile.py [L=32] [chi=256] [n_sweeps=12] [profiled=2]"""
import cProfile, io, pstats, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import toy_dmrg as td
from cyten_amd.block_backend import HipBlockBackend

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_sw = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n_prof = int(sys.argv[4]) if len(sys.argv) > 4 else 2
bb = HipBlockBackend('cuda:0')
td.dmrg(bb, td.heisenberg_model(L, 1.0), chi_max=chi, svd_min=1e-30, n_sweeps=1, lanczos_options=dict(N_max=4))
pr = cProfile.Profile()
want = '--no-profile' not in sys.argv


def on_sweep(k):
    if k == 1 and '--gc-freeze' in sys.argv:   # host runtime setting of the application: long-lived objects leave the collector
        import gc
        gc.collect()
        gc.freeze()
    if want and k == n_sw - n_prof:
        pr.enable()


stats = {}
E, psi, ts = td.dmrg(bb, td.heisenberg_model(L, 1.0), chi_max=chi, svd_min=1e-30, n_sweeps=n_sw, lanczos_options=dict(N_max=6),
                     sweep_times=True, stats=stats, on_sweep=on_sweep)
pr.disable()
bonds = 2 * (L - 1)
print(f'[dmrg-profile] L={L} chi={chi}: sweeps {[round(t, 3) for t in ts]} s; last {n_prof}: {sum(ts[-n_prof:]) / n_prof:.3f} s/sweep = '
      f'{1e3 * sum(ts[-n_prof:]) / n_prof / bonds:.2f} ms per bond update ({bonds} per sweep); E/L {E / L:.10f}; {stats}')
if want:
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(45)
    print(out.getvalue()[:9000])
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(30)
    print(out.getvalue()[:7000])
