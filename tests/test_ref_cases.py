"""CPU: the oracle reproduces every literal case the reference's own tests hold for the block-backend layer
(tests/golden/ref_block_backend_cases.json, transcribed as data with file:line provenance).  This is what PINS the
oracle to reference-held data ('oracle unpinned'); the same cases run through the HIP path in
tests/test_gpu_ref_cases.py."""
import numpy as np
import pytest

from oracle import block_ops as ops
from ref_cases import load_cases, run_case


class OracleApi:
    """numpy values play the role of blocks; 0-d numpy values the role of Scalars"""

    def block(self, a):
        return np.array(a)

    def zeros(self, shape):
        return np.zeros(shape)

    def shape(self, a):
        return np.shape(a)

    def dtype_name(self, a):
        return np.asarray(a).dtype.name

    def sum_all(self, a):
        return float(np.sum(a))

    def to_numpy(self, a):
        return np.asarray(a)

    def copy_block(self, a):
        return np.array(a, copy=True)

    def getitem(self, a, key):
        return ops.get_item(a, key)

    def setitem(self, a, key, value):
        return ops.set_item(a, key, value)

    def is_scalar(self, x):
        return np.ndim(x) == 0

    def scalar(self, v):
        return np.asarray(v)[()]

    def scalar_value(self, s):
        return s.item() if hasattr(s, 'item') else s

    def abs(self, a):
        return ops.abs_block(a)

    def scalar_unary(self, fn, z):
        return ops.scalar_unary(fn, z)

    def scalar_pow(self, z, e):
        return ops.scalar_pow(z, e)

    def apply_leg_permutations(self, a, perms):
        return ops.apply_leg_permutations(a, perms)

    def argmin(self, a):
        return ops.argmin(a)

    def matrix_exp(self, a):
        return ops.matrix_exp(a)

    def outer(self, a, b):
        return ops.outer(a, b)

    def kron(self, a, b):
        return ops.kron(a, b)

    def tdot(self, a, b, ia, ib):
        return ops.tdot(a, b, ia, ib)


@pytest.mark.parametrize('case', load_cases(), ids=lambda c: c['id'])
def test_oracle_reproduces_reference_held_case(case):
    run_case(OracleApi(), case)


def test_every_case_names_its_reference_test():
    for c in load_cases():
        assert 'tests/python_tests/' in c['provenance'] and ':' in c['provenance'], c['id']
