"""End-to-end known answer: the reference's ``test_dmrg_tfi`` (tests/python_tests/test_toycodes.py:92-105) -- two-site
DMRG with Z2 conservation on an open transverse-field Ising chain against exact diagonalisation, |dE| < 1e-9 -- run on
the block-sparse path of this repo (tests/toy_dmrg.py): compose, leg rotations, Lanczos on the two-site effective
Hamiltonian, combine_legs + batched SVD + truncation + split_legs, environment updates.  CPU: the numpy stand-in backend
(host logic); GPU: the HIP backend through the C-ABI."""
import numpy as np
import pytest

import toy_dmrg as td
from numpy_backend import NumpyGroupedBackend


@pytest.mark.parametrize('seed', [0, 1])
def test_tfi_ground_state_energy_host_logic(seed):
    J, g = np.random.default_rng(seed).random(2)             # the reference draws J, g the same way (np_random.random(2))
    L = 10
    E, psi = td.dmrg(NumpyGroupedBackend(), td.tfi_model(L, J, g), chi_max=32, n_sweeps=4)
    assert abs(E - td.tfi_exact_energy(L, J, g)) < 1e-9
    assert max(t.legs[2].dim for t in psi) <= 32


def test_tfi_charges_and_truncation_host_logic():
    """A truncated run stays variational (E >= E_exact) and close; every tensor obeys the charge rule."""
    L, J, g = 10, 1.0, 1.0                                    # critical point: the entanglement needs the bond dimension
    E_exact = td.tfi_exact_energy(L, J, g)
    E4, psi = td.dmrg(NumpyGroupedBackend(), td.tfi_model(L, J, g), chi_max=4, n_sweeps=3)
    for t in psi:
        t.check_charges()
        assert t.legs[2].dim <= 4
    assert E4 >= E_exact - 1e-12 and E4 - E_exact < 1e-3


@pytest.mark.gpu
def test_tfi_ground_state_energy_on_device(bb):
    J, g = np.random.default_rng(2).random(2)
    L = 10
    E, psi = td.dmrg(bb, td.tfi_model(L, J, g), chi_max=32, n_sweeps=3, lanczos_options=dict(N_max=20))
    assert abs(E - td.tfi_exact_energy(L, J, g)) < 1e-9
    for t in psi:
        t.check_charges()


def test_heisenberg_u1_ground_state_energy_host_logic():
    """U(1) (2 Sz) conservation with charged MPO bonds (s+ / s- hopping), Neel initial state: the known answer of the
    reference's test_dmrg_heisenberg (test_toycodes.py:58-71: exact diagonalisation, 1e-9)."""
    L, J = 8, 1.0
    E, psi = td.dmrg(NumpyGroupedBackend(), td.heisenberg_model(L, J), chi_max=40, n_sweeps=4)
    assert abs(E - td.heisenberg_exact_energy(L, J)) < 1e-9
    for t in psi:
        t.check_charges()
    assert max(t.legs[2].nsec for t in psi) >= 3          # several charge sectors on the middle bonds


@pytest.mark.gpu
def test_heisenberg_u1_ground_state_energy_on_device(bb):
    L, J = 8, 1.0
    E, psi = td.dmrg(bb, td.heisenberg_model(L, J), chi_max=40, n_sweeps=3, lanczos_options=dict(N_max=20))
    assert abs(E - td.heisenberg_exact_energy(L, J)) < 1e-9
    from cyten_amd.block_backend import HipBlock
    assert all(isinstance(b, HipBlock) for t in psi for b in t.blocks)      # the state lives on the device
