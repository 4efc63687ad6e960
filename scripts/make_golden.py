"""Generate the golden fixtures under tests/golden/ with the CPU oracle.

The reference's own tests hold no golden vectors for this path; these
fixtures are inputs + expected outputs produced by `oracle/` (the same numpy/scipy routines the
reference calls, driven by the restated host loops) so that the GPU box can compare the HIP path
against committed data even without re-running the oracle.  Small on purpose (a few hundred KB).

Usage:  python scripts/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cyten_amd import workloads as wl  # noqa: E402
from oracle import abelian_ref as ref  # noqa: E402
from oracle import block_ops as ops  # noqa: E402
from oracle import krylov_ref  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def spec_arrays(prefix, t):
    d = {f'{prefix}_moduli': np.array(t.moduli, dtype=np.int64), f'{prefix}_block_inds': t.block_inds,
         f'{prefix}_num_codomain': np.array(t.num_codomain), f'{prefix}_nlegs': np.array(len(t.legs))}
    for k, l in enumerate(t.legs):
        d[f'{prefix}_leg{k}_sectors'] = l.sectors
        d[f'{prefix}_leg{k}_mults'] = l.mults
        d[f'{prefix}_leg{k}_sign'] = np.array(l.sign)
    for i, b in enumerate(t.blocks):
        d[f'{prefix}_block{i}'] = b
    return d


def theta_fixture(name, A, B, chi_max):
    res = ref.theta_tdot_svd(A, B, chi_max=chi_max)
    d = {}
    d.update(spec_arrays('A', A))
    d.update(spec_arrays('B', B))
    d['theta_block_inds'] = res['theta_block_inds']
    for i, b in enumerate(res['theta_blocks']):
        d[f'theta_block{i}'] = b
    d['charges'] = np.array(res['charges'], dtype=np.int64)
    d['S_all'] = res['S_all']
    d['S_sizes'] = np.array([len(s) for _, s, _ in res['usv']])
    d['mask'] = res['mask']
    d['err'] = np.array(res['err'])
    d['new_norm'] = np.array(res['new_norm'])
    d['n_matrix_dot'] = np.array(res['n_matrix_dot'])
    d['chi_max'] = np.array(chi_max)
    np.savez_compressed(os.path.join(OUT, name), **d)


def decomposition_fixture():
    rng = np.random.default_rng(2024)
    d = {}
    mats = [rng.standard_normal((9, 6)), rng.standard_normal((6, 9)), rng.standard_normal((20, 4)) @ rng.standard_normal((4, 17)),
            rng.standard_normal((33, 33)) * np.logspace(0, -8, 33)[None, :]]
    for i, m in enumerate(mats):
        d[f'svd_in{i}'] = m
        d[f'svd_S{i}'] = ops.matrix_svd(m)[1]
        q, r = ops.matrix_qr(m, False)
        d[f'qr_R{i}'] = r
    for i, n in enumerate([5, 31]):
        h = rng.standard_normal((n, n))
        h = (h + h.T) / 2
        d[f'eigh_in{i}'] = h
        d[f'eigh_W{i}'] = ops.eigvalsh(h)
    d['n_svd'] = np.array(len(mats))
    d['n_eigh'] = np.array(2)
    np.savez_compressed(os.path.join(OUT, 'decompositions.npz'), **d)


def heff_fixture():
    """Two-site effective Hamiltonian: tensors, H_eff theta and the Lanczos ground state of
    the dense oracle (reference defaults except N_max / reortho, stated in the fixture)."""
    cfg = wl.config_heff(40, 3, seed=21)
    dense = {k: ref.to_dense(v) for k, v in cfg.items()}
    mv = krylov_ref.heff_dense(dense['LP'], dense['W1'], dense['W2'], dense['RP'])
    d = {}
    for k, v in cfg.items():
        d.update(spec_arrays(k, v))
    d['matvec_dense'] = mv(dense['theta'])
    E0, psi, N = krylov_ref.lanczos_dense(mv, dense['theta'], N_max=30, reortho=True)
    d['E0'] = np.array(E0)
    d['psi_dense'] = psi
    d['N'] = np.array(N)
    d['N_max'] = np.array(30)
    np.savez_compressed(os.path.join(OUT, 'heff_u1_chi40.npz'), **d)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    theta_fixture('theta_z2_chi64.npz', *wl.config_z2_chi64(), chi_max=40)
    theta_fixture('theta_u1_chi96.npz', *wl.config_u1_mps(96), chi_max=60)
    theta_fixture('theta_u1u1_chi120.npz', *wl.config_u1u1_mps(120), chi_max=100)
    decomposition_fixture()
    heff_fixture()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
