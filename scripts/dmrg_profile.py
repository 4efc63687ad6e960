"""cProfile of one toy-DMRG sweep on the HIP backend (development aid)."""
import sys, cProfile, pstats
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import toy_dmrg as td
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
model = td.heisenberg_model(32, 1.0)
st = {}
E, psi, ts = td.dmrg(bb, model, chi_max=128, svd_min=1e-30, n_sweeps=9, lanczos_options=dict(N_max=6), sweep_times=True, stats=st)
print('sweep times', [round(t, 2) for t in ts], st, 'E', E)
pr = cProfile.Profile(); pr.enable()
E, psi, ts = td.dmrg(bb, model, chi_max=128, svd_min=1e-30, n_sweeps=9, lanczos_options=dict(N_max=6), sweep_times=True)
pr.disable()
print('sweep times (profiled)', [round(t, 2) for t in ts])
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
