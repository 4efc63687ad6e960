"""Device side of rows a11 / f4 against the reference-held tree-move expectations (tests/golden/ref_tree_move_cases.json,
from test_fusion_tree_backend.py:36-188, :401-617, :634-786): `HipBlockBackend.transform_blocks` -- one zero fill and ONE
`cyb_lincomb_strided_batched_c128` launch per tensor, complex coefficients -- through the C-ABI."""
import numpy as np
import pytest

from oracle import block_ops as ops
from tree_move_fixture import expected, inputs, load, updates

pytestmark = pytest.mark.gpu
CASES, SYM = load()


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_device_transform_blocks_reproduces_the_reference_expectation(bb, case, rng):
    old = inputs(case, rng)
    want, mask = expected(case, SYM, old)
    ups = updates(case, SYM)
    shapes = [tuple(s) for s in case['new_shapes']]
    dev_old = [bb.as_block(o) for o in old]
    got = [bb.to_numpy(g) for g in bb.transform_blocks(dev_old, shapes, ups)]
    ref = ops.transform_blocks(old, shapes, ups)
    for g, w, m, r in zip(got, want, mask, ref):
        assert g.dtype == np.complex128 and g.shape == w.shape
        assert np.abs(g - w)[m].max(initial=0.0) <= 1e-14
        assert np.abs(g[~m]).max(initial=0.0) == 0.0
        assert np.abs(g - r).max(initial=0.0) <= 1e-14
    # the general route (view objects, `lincomb_many`) on old blocks that are no plain row-major matrices: transposed storage
    views = [bb.permute_axes(bb.as_block(np.ascontiguousarray(o.T)), [1, 0]) for o in old]
    assert all(v.strides[1] != 1 or min(v.shape) == 1 for v in views)
    got2 = [bb.to_numpy(g) for g in bb.transform_blocks(views, shapes, ups)]
    for g, r in zip(got2, ref):
        assert np.abs(g - r).max(initial=0.0) <= 1e-14


@pytest.mark.parametrize('case', [c for c in CASES if c['name'].startswith('fib_c') or c['name'] == 'fib_b_bend_up'], ids=lambda c: c['name'])
def test_device_real_data_under_a_complex_mapping(bb, case, rng):
    """float64 blocks with complex symbols: the result is complex128 (fusion_tree_mapping.cpp:433-436); float64 sources are
    read as they are (`src_real` terms), no complex copy of the tensor is made first."""
    old = inputs(case, rng, real=True)
    want, mask = expected(case, SYM, old)
    got = bb.transform_blocks([bb.as_block(o) for o in old], [tuple(s) for s in case['new_shapes']], updates(case, SYM))
    cplx = any(abs(SYM[t['coeff']].imag) > 0 for st in case['statements'] for t in st['terms'])
    for g, w, m in zip(got, want, mask):
        assert g.is_complex == cplx          # (the B symbols of the Fibonacci category are real: the data stays float64)
        assert np.abs(bb.to_numpy(g) - w)[m].max(initial=0.0) <= 1e-14


def test_device_lincomb_complex_views(bb, rng):
    """`lincomb_many` on complex views: permuted sources, accumulate, mixed real / complex sources, 5 axes"""
    a = rng.standard_normal((3, 4, 2, 5, 2)) + 1j * rng.standard_normal((3, 4, 2, 5, 2))
    b = rng.standard_normal((2, 5, 2, 3, 4)) + 1j * rng.standard_normal((2, 5, 2, 3, 4))
    r = rng.standard_normal((4, 3, 2, 5, 2))
    d0 = rng.standard_normal(a.shape) + 1j * rng.standard_normal(a.shape)
    A, B, R, D = bb.as_block(a), bb.as_block(b), bb.as_block(r), bb.as_block(d0.copy())
    ca, cb, cr = 0.3 - 1.2j, -2.0 + 0.5j, 1.5j
    bb.lincomb_many([(D, [(ca, A), (cb, bb.permute_axes(B, [3, 4, 0, 1, 2])), (cr, bb.permute_axes(R, [1, 0, 2, 3, 4]))], True)])
    want = d0 + ca * a + cb * np.transpose(b, [3, 4, 0, 1, 2]) + cr * np.transpose(r, [1, 0, 2, 3, 4])
    assert np.abs(bb.to_numpy(D) - want).max() <= 1e-13
    with pytest.raises(NotImplementedError):
        bb.lincomb_many([(R, [(1j, R)], False)])          # a float64 destination cannot take a complex coefficient
