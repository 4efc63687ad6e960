"""Driver for the reference-held literal cases of tests/golden/ref_block_backend_cases.json: one `run_case(api, case)`
used by the CPU test (api = the numpy oracle) and the GPU test (api = the HIP backend through the C-ABI).

`api` is a small adapter object; see OracleApi (tests/test_ref_cases.py) and HipApi (tests/test_gpu_ref_cases.py)."""
import json
import os

import numpy as np
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    with open(os.path.join(HERE, 'golden', 'ref_block_backend_cases.json')) as f:
        return json.load(f)['cases']


def _key(k):
    """JSON key -> python index: "1:3" / ":" -> slice, list -> index array, int -> int"""
    out = []
    for x in k:
        if isinstance(x, str):
            parts = [int(p) if p else None for p in x.split(':')]
            out.append(slice(*parts))
        elif isinstance(x, list):
            out.append(list(x))
        else:
            out.append(int(x))
    return tuple(out)


def _input(case):
    if 'input_arange' in case:
        shp = case['input_arange']
        return np.arange(int(np.prod(shp)), dtype=np.float64).reshape(shp)
    if 'input_complex' in case:
        return np.array([complex(r, i) for r, i in case['input_complex']])
    return np.array(case['input'], dtype=np.float64)


def _expect(case, inp=None):
    if 'expect_from' in case:   # the reference test computes its expectation with this library call
        return {'numpy.exp(4.0)': lambda: np.exp(4.0), 'numpy.log(4.0)': lambda: np.log(4.0),
                'scipy.linalg.expm(input)': lambda: scipy.linalg.expm(inp)}[case['expect_from']]()
    return np.array(case['expect']) if isinstance(case['expect'], list) else case['expect']


def run_case(api, case):
    """Runs one case through `api`; asserts the reference's expectation (bit-exact unless the case carries rtol/atol)."""
    op = case['op']
    exact = 'rtol' not in case

    def check(got, want):
        got, want = np.asarray(got), np.asarray(want)
        assert got.shape == want.shape, (case['id'], got.shape, want.shape)
        if exact:
            np.testing.assert_array_equal(got, want, err_msg=case['id'])
        else:
            np.testing.assert_allclose(got, want, rtol=case['rtol'], atol=case['atol'], err_msg=case['id'])

    if op == 'zeros':
        z = api.zeros(case['shape'])
        assert tuple(api.shape(z)) == tuple(case['expect_shape']) and api.dtype_name(z) == case['expect_dtype']
        assert api.sum_all(z) == case['expect_sum']
    elif op == 'copy_block':
        check(api.to_numpy(api.copy_block(api.block(_input(case)))), _expect(case))
    elif op == 'getitem_scalar':
        blk = api.block(_input(case))
        k = case['key']
        key = tuple(k) if case['key_kind'] == 'tuple' else list(k) if case['key_kind'] == 'list' else int(k[0])
        s = api.getitem(blk, key)
        assert api.is_scalar(s), case['id']
        assert api.scalar_value(s) == case['expect'], case['id']
    elif op == 'setitem_scalar':
        blk = api.block(_input(case))
        k = tuple(case['key']) if len(case['key']) > 1 else int(case['key'][0])
        blk = api.setitem(blk, k, api.scalar(case['value']))
        assert api.scalar_value(api.getitem(blk, k)) == case['value'], case['id']
        check(api.to_numpy(blk), np.array(case['expect_after']))
    elif op == 'getitem_block':
        got = api.getitem(api.block(_input(case)), _key(case['key']))
        assert not api.is_scalar(got), case['id']
        check(api.to_numpy(got), _expect(case))
    elif op == 'setitem_block':
        blk = api.setitem(api.block(_input(case)), _key(case['key']), api.block(np.array(case['value'])))
        check(api.to_numpy(blk), np.array(case['expect_after']))
    elif op == 'abs':
        out = api.abs(api.block(_input(case)))
        check(api.to_numpy(out), _expect(case))
    elif op == 'scalar_unary':
        re, im = case['value']
        z = api.scalar(complex(re, im) if im != 0.0 else re)
        out = api.scalar_unary(case['fn'], z)
        assert api.is_scalar(out), case['id']
        assert api.scalar_value(out) == _expect(case), case['id']
        if 'expect_dtype' in case:
            assert api.dtype_name(out) == case['expect_dtype'], case['id']
    elif op == 'scalar_pow':
        e = api.scalar(case['exponent']) if case['exponent_kind'] == 'scalar' else case['exponent']
        out = api.scalar_pow(api.scalar(case['value']), e)
        assert api.is_scalar(out) and api.scalar_value(out) == case['expect'], case['id']
    elif op == 'apply_leg_permutations':
        perms = [np.array(p, dtype=np.int64) for p in case['perms']]
        out = api.apply_leg_permutations(api.block(_input(case)), perms)
        check(api.to_numpy(out), _expect(case))
    elif op == 'argmin':
        assert tuple(int(i) for i in api.argmin(api.block(_input(case)))) == tuple(case['expect']), case['id']
    elif op == 'matrix_exp':
        inp = _input(case)
        check(api.to_numpy(api.matrix_exp(api.block(inp))), _expect(case, inp))
    elif op in ('outer', 'kron'):
        out = getattr(api, op)(api.block(np.array(case['a'])), api.block(np.array(case['b'])))
        if 'expect_shape' in case:
            assert tuple(api.shape(out)) == tuple(case['expect_shape']), case['id']
        check(api.to_numpy(out), _expect(case))
    elif op == 'tdot':
        out = api.tdot(api.block(np.array(case['a'])), api.block(np.array(case['b'])), case['idcs_a'], case['idcs_b'])
        assert tuple(api.shape(out)) == tuple(case['expect_shape']), case['id']
        check(api.to_numpy(out), _expect(case))
    else:
        raise AssertionError(f'unknown op {op} in case {case["id"]}')
