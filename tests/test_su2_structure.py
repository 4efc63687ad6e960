"""cfg4 / row f4 on REAL non-abelian structure, CPU part: the committed SU(2) and SU(2) x U(1) fixture
(fusion rules + Wigner 6j symbols from sympy, scripts/make_su2_golden.py) is internally consistent and the oracle's
``transform_blocks`` -- the block arithmetic of TreePairMapping::transform_tensor, fusion_tree_mapping.cpp:391-513 --
applies the F-move as the recoupling theory says: orthogonal per multiplicity space, so old -> new -> old is the identity."""
import numpy as np
import pytest

from oracle import block_ops as ops
from su2_fixture import compose_lists, load, tree_move


def test_compose_lists_follow_the_fusion_rules():
    z = load()
    lists = compose_lists(z)
    assert len(lists['su2']) >= 4 and len(lists['su2xu1']) >= 25
    # spin-1/2 chain: rows(J) = m_L(J - 1/2) + m_L(J + 1/2) with the stored multiplicities
    mL = dict(zip(z['L2'].tolist(), z['mL'].tolist()))
    for (J2, rows, K, cols) in z['compose'].tolist():
        assert rows == mL.get(J2 - 1, 0) + mL.get(J2 + 1, 0) == cols
        assert K == dict(zip(z['M2'].tolist(), z['mM'].tolist()))[J2]
    # every bond carries sum_j m_j (2j + 1) ~ chi states
    assert abs(int(np.sum(z['mL'] * (z['L2'] + 1))) - 512) < 64


@pytest.mark.parametrize('which', ['su2', 'su2xu1'])
def test_f_move_is_orthogonal_and_round_trips_through_the_oracle(which, rng):
    z = load()
    keys, rows, ncols, fwd, inv = tree_move(z, which)
    assert len(fwd) == len(inv) and (len(fwd) >= 15 if which == 'su2' else len(fwd) >= 250)
    old = [rng.standard_normal((r, c)) for r, c in zip(rows, ncols)]
    shapes = [(r, c) for r, c in zip(rows, ncols)]
    new = ops.transform_blocks(old, shapes, fwd)
    back = ops.transform_blocks(new, shapes, inv)
    for a, b, n in zip(old, back, new):
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-13 * max(1.0, np.abs(a).max()))       # F^T F = 1
        assert abs(np.linalg.norm(n) - np.linalg.norm(a)) <= 1e-12 * np.linalg.norm(a)         # an orthogonal change of basis
    # dense check of one coupled sector: the map on its rows is the block matrix built from the stored coefficients
    b0 = max(range(len(rows)), key=lambda k: rows[k])
    Fbig = np.zeros((rows[b0], rows[b0]))
    for (b, (r0, r1), _, _, _, _, _, terms) in fwd:
        if b == b0:
            for c, k, (o0, o1), _ in terms:
                Fbig[r0:r1, o0:o1] += c * np.eye(r1 - r0)
    np.testing.assert_allclose(Fbig @ Fbig.T, np.eye(rows[b0]), atol=1e-12)
    np.testing.assert_allclose(new[b0], Fbig @ old[b0], atol=1e-12 * np.abs(old[b0]).max())
