"""cfg4 / row f4 on real SU(2) and SU(2) x U(1) block structure, GPU part: the coupled-sector GEMM list of a chi = 512
two-site compose (FusionTreeBackend::compose, fusion_tree_backend.cpp:669-698: one matrix_dot per coupled sector) through
the grouped launch, the per-sector SVD list (fusion_tree_backend.cpp:2184-2252) through the batched call, and the F-move of
TreePairMapping::transform_tensor (fusion_tree_mapping.cpp:391-513) through ``transform_blocks`` -- against numpy / the oracle."""
import numpy as np
import pytest

from oracle import block_ops as ops
from su2_fixture import compose_lists, load, tree_move

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.mark.parametrize('which', ['su2', 'su2xu1'])
def test_coupled_sector_gemm_and_svd_lists(bb, rng, which):
    shapes = compose_lists(load())[which]
    A = [rng.standard_normal((r, k)) for r, k, _ in shapes]
    B = [rng.standard_normal((k, c)) for _, k, c in shapes]
    outs = bb.matrix_dot_grouped([[(bb.as_block(a), bb.as_block(b))] for a, b in zip(A, B)])      # ONE launch for all sectors
    want = [a @ b for a, b in zip(A, B)]
    for o, w in zip(outs, want):
        np.testing.assert_allclose(bb.to_numpy(o), w, rtol=0, atol=TOL * max(1.0, np.abs(w).max()))
    res = bb.matrix_svd_batched(outs)                                                               # ONE batched call
    for (u, s, vh), w in zip(res, want):
        u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
        nrm = max(np.linalg.norm(w), 1e-300)
        assert np.abs(s - np.linalg.svd(w, compute_uv=False)).max() <= TOL * nrm
        assert np.abs((u * s) @ vh - w).max() <= TOL * nrm
        assert np.abs(u.T @ u - np.eye(len(s))).max() <= TOL and np.abs(vh @ vh.T - np.eye(len(s))).max() <= TOL


@pytest.mark.parametrize('which', ['su2', 'su2xu1'])
def test_f_move_on_the_device_matches_the_oracle_and_round_trips(bb, rng, which):
    keys, rows, ncols, fwd, inv = tree_move(load(), which)
    shapes = [(r, c) for r, c in zip(rows, ncols)]
    old = [rng.standard_normal(sh) for sh in shapes]
    want = ops.transform_blocks(old, shapes, fwd)
    dev_old = [bb.as_block(a) for a in old]
    got = bb.transform_blocks(dev_old, shapes, fwd)                   # one zero-fill + ONE launch for every tree block
    for g, w in zip(got, want):
        np.testing.assert_allclose(bb.to_numpy(g), w, rtol=0, atol=1e-13 * max(1.0, np.abs(w).max()))
    back = bb.transform_blocks(got, shapes, inv)
    for b, a in zip(back, old):
        np.testing.assert_allclose(bb.to_numpy(b), a, rtol=0, atol=1e-12 * max(1.0, np.abs(a).max()))
