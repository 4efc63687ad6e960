"""Rows a11 / f4 pinned by what the reference holds: the literal expected blocks of single C-symbol and B-symbol moves in
tests/python_tests/backends/test_fusion_tree_backend.py (Fibonacci :36-188, :634-786; SU(3)_3 :401-617) against the oracle's
restatement of the block arithmetic of TreePairMapping::transform_tensor (oracle.block_ops.transform_blocks,
fusion_tree_mapping.cpp:433-507).  CPU, every round; tests/test_gpu_tree_moves.py runs the device path against the same file."""
import numpy as np
import pytest

from oracle import block_ops as ops
from tree_move_fixture import expected, inputs, load, updates

CASES, SYM = load()


def test_fixture_symbols_are_the_closed_forms_the_reference_states():
    phi = (1 + 5 ** 0.5) / 2
    assert abs(SYM['R_1'] - np.exp(-4j * np.pi / 5)) < 1e-15 and abs(SYM['R_tau'] - np.exp(3j * np.pi / 5)) < 1e-15
    # the reference prints the values it expects next to the formulas (test_fusion_tree_backend.py:72-73)
    assert abs(SYM['R_1'] - (-0.8090 - 0.5878j)) < 1e-4 and abs(SYM['R_tau'] - (-0.3090 + 0.9511j)) < 1e-4
    assert abs(SYM['C_tttt11'] - (-0.5000 + 0.3633j)) < 1e-4 and abs(SYM['C_tttt1t'] - (-0.2429 - 0.7477j)) < 1e-4
    assert abs(SYM['C_ttttt1'] - (-0.2429 - 0.7477j)) < 1e-4 and abs(SYM['C_tttttt'] - (-0.6180)) < 1e-4
    assert abs(SYM['C_tttttt'] + 1 / phi) < 1e-15
    # unitarity of the 2 x 2 mixing blocks the expectations are built from (C and F moves are unitary)
    c = np.array([[SYM['C_tttt11'], SYM['C_ttttt1']], [SYM['C_tttt1t'], SYM['C_tttttt']]])
    assert np.abs(c @ c.conj().T - np.eye(2)).max() < 1e-15
    for k in ('f2f2', 'f1f1', 'f1rf2', 'f2rf1'):
        m = np.array([[SYM[f'{k}_00'], SYM[f'{k}_10']], [SYM[f'{k}_01'], SYM[f'{k}_11']]])
        assert np.abs(m @ m.conj().T - np.eye(2)).max() < 1e-15


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_oracle_transform_blocks_reproduces_the_reference_expectation(case, rng):
    old = inputs(case, rng)
    want, mask = expected(case, SYM, old)
    got = ops.transform_blocks(old, [tuple(s) for s in case['new_shapes']], updates(case, SYM))
    for g, w, m in zip(got, want, mask):
        assert g.shape == w.shape and g.dtype == np.complex128
        assert np.abs(g - w)[m].max(initial=0.0) <= 1e-14          # the reference's own tolerance (eps = 1e-14, :40)
        assert np.abs(g[~m]).max(initial=0.0) == 0.0               # nothing outside the written-out rows / columns
    if 'partial' not in case and case['axis'] != 'element':
        # a braid is unitary: the move preserves the norm of every coupled-sector block (:192-196 check it through repeated braids)
        for g, o in zip(got, old):
            assert abs(np.linalg.norm(g) - np.linalg.norm(o)) <= 1e-13 * np.linalg.norm(o)


@pytest.mark.parametrize('case', [c for c in CASES if c['name'].startswith('fib_c')], ids=lambda c: c['name'])
def test_real_data_under_a_complex_mapping_becomes_complex(case, rng):
    """dtype = to_complex(dtype) when the mapping is not real (fusion_tree_mapping.cpp:433-436)"""
    old = inputs(case, rng, real=True)
    want, _ = expected(case, SYM, old)
    got = ops.transform_blocks(old, [tuple(s) for s in case['new_shapes']], updates(case, SYM))
    for g, w in zip(got, want):
        assert g.dtype == np.complex128 and np.abs(g - w).max() <= 1e-14
