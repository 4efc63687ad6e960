"""GPU: the HIP path (through the C-ABI) reproduces every literal case the reference's own tests hold for the
block-backend layer -- bit-exact for the data-movement / indexing / exactly-representable ones, to the reference test's
own tolerance where it states one (matrix_exp: rtol = atol = 1e-12)."""
import numpy as np
import pytest

from cyten_amd.block_backend import HipBlock, Scalar
from ref_cases import load_cases, run_case

pytestmark = pytest.mark.gpu


class HipApi:
    def __init__(self, bb):
        self.bb = bb

    def block(self, a):
        return self.bb.as_block(np.array(a))

    def zeros(self, shape):
        return self.bb.zeros(shape, dtype='float64')

    def shape(self, a):
        return self.bb.get_shape(a)

    def dtype_name(self, a):
        return np.dtype(a.dtype).name

    def sum_all(self, a):
        return float(self.bb.sum_all(a))

    def to_numpy(self, a):
        return self.bb.to_numpy(a)

    def copy_block(self, a):
        return self.bb.copy_block(a)

    def getitem(self, a, key):
        return a[key]

    def setitem(self, a, key, value):
        a[key] = value
        return a

    def is_scalar(self, x):
        assert isinstance(x, (Scalar, HipBlock))
        return isinstance(x, Scalar)

    def scalar(self, v):
        return self.bb.as_scalar(v)

    def scalar_value(self, s):
        return s.as_float64()

    def abs(self, a):
        return abs(a)

    def scalar_unary(self, fn, z):
        return abs(z) if fn == 'abs' else getattr(z, fn)()

    def scalar_pow(self, z, e):
        return z ** e if not isinstance(e, Scalar) else z.pow(e)

    def apply_leg_permutations(self, a, perms):
        return self.bb.apply_leg_permutations(a, perms)

    def argmin(self, a):
        return self.bb.argmin(a)

    def matrix_exp(self, a):
        return self.bb.matrix_exp(a)

    def outer(self, a, b):
        return self.bb.outer(a, b)

    def kron(self, a, b):
        return self.bb.kron(a, b)

    def tdot(self, a, b, ia, ib):
        return self.bb.tdot(a, b, ia, ib)


@pytest.mark.parametrize('case', load_cases(), ids=lambda c: c['id'])
def test_hip_reproduces_reference_held_case(bb, case):
    run_case(HipApi(bb), case)


def test_scalar_is_a_device_value_with_the_reference_accessors(bb):
    """BlockBackend::Scalar (block_backend.h:170-240): accessors raise on the wrong dtype exactly like block_backend.cpp:284-330"""
    s = bb.as_scalar(3.0 + 4.0j)
    assert s.dtype == np.dtype('complex128') and s.as_complex128() == 3 + 4j
    with pytest.raises(RuntimeError):
        s.as_float64()
    with pytest.raises(RuntimeError):
        bb.as_scalar(1.0).as_bool()
    with pytest.raises(RuntimeError):
        bb.as_scalar(1.0).as_int64()
    assert bb.as_scalar(True).as_bool() is True
    t = bb.as_scalar(2.0)
    assert (t + 1.0).as_float64() == 3.0 and (t - bb.as_scalar(0.5)).as_float64() == 1.5 and (-t).as_float64() == -2.0
    assert (t * 4.0).as_float64() == 8.0 and (t / 4.0).as_float64() == 0.5 and t.inverse().as_float64() == 0.5
    assert (t < 3.0).as_bool() and not (t >= 3.0).as_bool() and (t == 2.0).as_bool() and (s != 1.0).as_bool()
    with pytest.raises(RuntimeError):
        bb.as_scalar(0.0).inverse()
    assert isinstance(t.to_numpy(), np.float64) and float(t) == 2.0 and complex(s) == 3 + 4j
    with pytest.raises(ValueError):
        Scalar(bb.zeros((2,)))


def test_set_item_writes_through_gather_keys(bb):
    """keys that get_item serves as copies (index arrays, boolean masks, negative steps) must change `a`."""
    rng = np.random.default_rng(3)
    for key, vshape in [(([2, 0], slice(None)), (2, 5)), ((slice(None), [4, 1, 1]), (4, 3)), ((slice(None, None, -1), slice(1, 4)), (4, 3)),
                        ((np.array([True, False, True, True]), slice(None)), (3, 5)), ((slice(3, 0, -2), 2), (2,)), ((1, [0, 3]), (2,))]:
        a = rng.standard_normal((4, 5))
        v = rng.standard_normal(vshape)
        want = a.copy()
        want[key] = v
        blk = bb.as_block(a)
        bb.set_item(blk, key, bb.as_block(v))
        np.testing.assert_array_equal(bb.to_numpy(blk), want)
    blk = bb.as_block(np.zeros((3, 3)))
    blk[[0, 2], :] = bb.as_scalar(7.0)        # a Scalar broadcasts like a number
    np.testing.assert_array_equal(bb.to_numpy(blk), np.array([[7.0] * 3, [0.0] * 3, [7.0] * 3]))
