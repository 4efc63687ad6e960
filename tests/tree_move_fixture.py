"""Shared reader of tests/golden/ref_tree_move_cases.json (the literal tree-move expectations the reference's tests hold,
test_fusion_tree_backend.py:36-188, :401-617, :634-786): seeded inputs, the expectation exactly as the reference writes it
(index lists and symbols), and the same move as `transform_blocks` updates -- one (tree-block pair, terms) record per
iteration of the loops of TreePairMapping::transform_tensor (fusion_tree_mapping.cpp:453-507)."""
import itertools
import json
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_tree_move_cases.json')


def load():
    with open(PATH) as f:
        doc = json.load(f)
    sym = {k: complex(v['re'], v['im']) for k, v in doc['symbols'].items()}
    return doc['cases'], sym


def inputs(case, rng, real=False):
    """random_uniform complex blocks (the reference draws its inputs the same way, :51-53); `real`: float64 blocks"""
    if real:
        return [rng.random(tuple(sh)) for sh in case['old_shapes']]
    return [rng.random(tuple(sh)) + 1j * rng.random(tuple(sh)) for sh in case['old_shapes']]


def expected(case, sym, old):
    """The expectation in the reference's own form: zero blocks, `expect[nb][idx, :] = sum blocks[ob][idx', :] * symbol`.
    Returns (blocks, masks): for a 'partial' case only the rows / columns the reference writes out are set in `masks`."""
    new = [np.zeros(tuple(sh), dtype=complex) for sh in case['new_shapes']]
    masks = [np.zeros(tuple(sh), dtype=bool) for sh in case['new_shapes']]
    for st in case['statements']:
        nb, dst = st['nb'], st['dst']
        for t in st['terms']:
            c, ob, src = sym[t['coeff']], t['ob'], t['src']
            if case['axis'] == 0:
                new[nb][dst, :] += c * old[ob][src, :]
            elif case['axis'] == 1:
                new[nb][:, dst] += c * old[ob][:, src]
            else:
                new[nb][dst[0], dst[1]] += c * old[ob][src[0], src[1]]
        if case['axis'] == 0:
            masks[nb][dst, :] = True
        elif case['axis'] == 1:
            masks[nb][:, dst] = True
        else:
            masks[nb][dst[0], dst[1]] = True
    return new, masks


def _axis_perm(dims, offsets):
    """permutation p of the multiplicity axes with  reshape(transpose(reshape(x, dims), p), -1) == x[offsets]"""
    n = int(np.prod(dims))
    ref = np.arange(n).reshape(dims)
    for p in itertools.permutations(range(len(dims))):
        if list(np.transpose(ref, p).reshape(-1)) == list(offsets):
            return list(p)
    raise ValueError(f'no axis permutation of {dims} gives {offsets}')


def updates(case, sym):
    """The move as `transform_blocks` updates (b, rows, cols, dims1, idcs1, dims2, idcs2, [(coeff, k, rows_k, cols_k)]): one
    record per (row tree, column tree) pair of the new block, as the reference's loops visit them."""
    ups = []
    if case['axis'] == 'element':
        for st in case['statements']:
            r, c = st['dst']
            terms = [(sym[t['coeff']], t['ob'], (t['src'][0], t['src'][0] + 1), (t['src'][1], t['src'][1] + 1)) for t in st['terms']]
            ups.append((st['nb'], (r, r + 1), (c, c + 1), [1], [0], [1], [1], terms))
        return ups
    rt, ct = case['row_tree'], case['col_tree']
    d1, d2 = rt['dims'], ct['dims']
    n1 = len(d1)
    ident1, ident2 = list(range(n1)), [n1 + i for i in range(len(d2))]
    for st in case['statements']:
        nb, dst = st['nb'], st['dst']
        if case['axis'] == 0:
            w = rt['width']
            ncols = case['new_shapes'][nb][1]
            for i0 in range(0, len(dst), w):
                drows = dst[i0:i0 + w]
                assert drows == list(range(drows[0], drows[0] + w)) and drows[0] % w == 0
                perm, srcs = None, []
                for t in st['terms']:
                    s = t['src'][i0:i0 + w]
                    base = s[0] - s[0] % w if w > 1 else s[0]
                    p = _axis_perm(d1, [x - base for x in s])
                    assert perm is None or perm == p
                    perm = p
                    srcs.append((sym[t['coeff']], t['ob'], base))
                for c0 in range(0, ncols, ct['width']):
                    cols = (c0, c0 + ct['width'])
                    ups.append((nb, (drows[0], drows[0] + w), cols, d1, perm, d2, ident2,
                                [(c, ob, (b, b + w), cols) for c, ob, b in srcs]))
        else:
            w = ct['width']
            nrows = case['new_shapes'][nb][0]
            for i0 in range(0, len(dst), w):
                dcols = dst[i0:i0 + w]
                assert dcols == list(range(dcols[0], dcols[0] + w)) and dcols[0] % w == 0
                perm, srcs = None, []
                for t in st['terms']:
                    s = t['src'][i0:i0 + w]
                    base = min(s)
                    assert base % w == 0
                    p = _axis_perm(d2, [x - base for x in s])
                    assert perm is None or perm == p
                    perm = p
                    srcs.append((sym[t['coeff']], t['ob'], base))
                for r0 in range(0, nrows, rt['width']):
                    rows = (r0, r0 + rt['width'])
                    ups.append((nb, rows, (dcols[0], dcols[0] + w), d1, ident1, d2, [n1 + i for i in perm],
                                [(c, ob, rows, (b, b + w)) for c, ob, b in srcs]))
    return ups
