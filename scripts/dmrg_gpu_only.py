"""The toy DMRG loop on the HIP backend only (for rocprofv3 kernel statistics of a whole run)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import toy_dmrg as td
from cyten_amd.block_backend import HipBlockBackend
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bb = HipBlockBackend('cuda:0')
E, psi, ts = td.dmrg(bb, td.heisenberg_model(32, 1.0), chi_max=chi, svd_min=1e-30, n_sweeps=11, lanczos_options=dict(N_max=6), sweep_times=True)
print('[dmrg-gpu] chi', chi, 'sweep times', [round(t, 3) for t in ts], 'E/L', E / 32, flush=True)
