"""The reference's one-block-at-a-time call pattern served by ONE grouped launch (cyten_amd/deferred.py):
replay of the contraction loop of abelian_compose_worker (abelian.cpp:1424-1460) through
DeferredBlockBackend, checked against the oracle and against the number of launches."""
import numpy as np
import pytest

from cyten_amd import abelian as ab
from cyten_amd import workloads as wl
from oracle import abelian_ref as ref

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope='module')
def dbb():
    from cyten_amd.deferred import DeferredBlockBackend
    return DeferredBlockBackend('cuda:0')


def test_reference_loop_becomes_one_launch(dbb):
    A, B = wl.config_u1_mps(256)
    a, b = ab.AbelianTensor.from_spec(dbb, A), ab.AbelianTensor.from_spec(dbb, B)
    plan = ab.compose_plan(a, b, 1)
    a2, b2 = ab._compose_operands(dbb, a, b, 1, plan)
    dbb.flush()
    f0, d0 = dbb.n_flushes, dbb.n_deferred
    res_blocks = []
    for pairs, shp in zip(plan.pairs, plan.res_shapes):     # the reference's loop, call for call
        i, j = pairs[0]
        block = dbb.matrix_dot(a2[i], b2[j])
        for i, j in pairs[1:]:
            block = block + dbb.matrix_dot(a2[i], b2[j])   # Block::operator+
        res_blocks.append(dbb.reshape(block, shp))
    n_dot = sum(len(p) for p in plan.pairs)
    assert dbb.n_deferred - d0 == n_dot and dbb.n_flushes == f0          # nothing launched yet
    assert all(tuple(blk.shape) == tuple(shp) for blk, shp in zip(res_blocks, plan.res_shapes))
    first = dbb.to_numpy(res_blocks[0])                                   # first observation: ONE launch for all
    assert dbb.n_flushes == f0 + 1
    blocks, bi, _ = ref.compose(A, B, 1)
    np.testing.assert_array_equal(plan.res_block_inds, bi)
    scale = max(np.abs(x).max() for x in blocks)
    assert np.abs(first - blocks[0]).max() <= TOL * scale
    for got, want in zip(res_blocks, blocks):
        assert np.abs(dbb.to_numpy(got) - want).max() <= TOL * scale
    assert dbb.n_flushes == f0 + 1                                        # and no further launch


def test_lazy_blocks_feed_other_kernels_and_chains(dbb, rng):
    a, b, c = (rng.standard_normal(s) for s in [(30, 20), (20, 25), (25, 10)])
    x = dbb.matrix_dot(dbb.as_block(a), dbb.as_block(b))             # pending
    y = dbb.matrix_dot(x, dbb.as_block(c))                           # depends on a pending product
    z = dbb.permute_axes(y, [1, 0])                                  # still metadata
    assert abs(dbb.norm(z) - np.linalg.norm(a @ b @ c)) <= TOL * np.linalg.norm(a @ b @ c)
    np.testing.assert_allclose(dbb.to_numpy(z), (a @ b @ c).T, atol=1e-11)
    u, s_, vh = dbb.matrix_svd(dbb.matrix_dot(dbb.as_block(a), dbb.as_block(b)))   # a decomposition observes
    np.testing.assert_allclose(dbb.to_numpy(s_), np.linalg.svd(a @ b, compute_uv=False), atol=1e-11)
    w = dbb.matrix_dot(dbb.as_block(a), dbb.as_block(b)) + dbb.as_block(np.ones((30, 25)))  # lazy + real block
    np.testing.assert_allclose(dbb.to_numpy(w), a @ b + 1.0, atol=1e-12)
    with pytest.raises(ValueError):
        dbb.matrix_dot(dbb.as_block(a), dbb.as_block(c))
