"""FusionTreeBackend callers on the CPU: the oracle's restatement of
``TreePairMapping::transform_tensor`` (oracle/fusion_tree_ref.py, fusion_tree_mapping.cpp:391-513) against (i) the literal
tree-move expectations the reference's tests hold (tests/golden/ref_tree_move_cases.json: Fibonacci C / B symbols, SU(3)_3 C
symbols) and (ii) dense leg permutations of abelian tensors with multiplicities > 1; the host logic of
``cyten_amd.fusion_tree`` (update records for ONE launch) on the numpy stand-in against the oracle; compose / svd / qr / lq /
eigh / truncation of FusionTreeData against their restatements and the invariants of the reference's tests."""
import numpy as np
import pytest

from cyten_amd import fusion_tree as ft
from fusion_tree_cases import AbelianTrees, spaces_from_fixture
from numpy_backend import NumpyGroupedBackend
from oracle import fusion_tree_ref as ref
from tree_move_fixture import expected, inputs, load

CASES, SYM = load()


class _NpBackend(NumpyGroupedBackend):
    """numpy stand-in + the two calls the fusion-tree callers add"""
    TRUNCATE_MAX = 0

    def transform_blocks(self, old_blocks, new_shapes, updates):
        from oracle import block_ops as ops
        return ops.transform_blocks(old_blocks, new_shapes, updates)

    def max_abs(self, a):
        return float(np.abs(a).max(initial=0.0))

    def as_block(self, a, dtype=None, device=None):
        return np.array(a)


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_transform_tensor_reproduces_the_reference_held_tree_moves(case, rng):
    old = inputs(case, rng)
    want, mask = expected(case, SYM, old)
    cod, dom, ncod, ndom, mapping = spaces_from_fixture(case, SYM)
    lg = case['legs']
    inds = [(b, b) for b in range(len(old))]
    blocks, rows = ref.transform_tensor(inds, old, cod, dom, ncod, ndom, lg['codomain_idcs'], lg['domain_idcs'], mapping)
    got = {tuple(r): b for r, b in zip(rows.tolist(), blocks)}
    for nb, (w, m) in enumerate(zip(want, mask)):
        if not m.any():
            assert (nb, nb) not in got
            continue
        g = got[(nb, nb)]
        assert np.abs(g - w)[m].max() <= 1e-14 and np.abs(g[~m]).max(initial=0.0) == 0.0
    # the host logic (one update list, one launch) against the statement-by-statement restatement
    res = ft.transform_tensor(_NpBackend(), ft.FusionTreeData(inds, old), cod, dom, ncod, ndom, lg['codomain_idcs'], lg['domain_idcs'], mapping)
    assert np.array_equal(res.block_inds, rows)
    for x, y in zip(res.blocks, blocks):
        assert np.abs(x - y).max(initial=0.0) <= 1e-15


@pytest.mark.parametrize('perm_c,perm_d', [((0, 1, 2), (0, 1)), ((1, 0, 2), (0, 1)), ((0, 2, 1), (1, 0)), ((2, 0, 1), (1, 0)), ((2, 1, 0), (0, 1))])
@pytest.mark.parametrize('cplx', [False, True])
def test_transform_tensor_is_the_dense_leg_permutation_for_abelian_trees(rng, perm_c, perm_d, cplx):
    at = AbelianTrees(rng)
    T = at.dense(rng, cplx)
    cod, dom, data = at.to_blocks(T, range(at.J), range(at.J, at.J + at.K))
    codomain_idcs, domain_idcs, ncf, ndf, mapping = at.braid(perm_c, perm_d)
    Tp = np.transpose(T, list(perm_c) + [at.J + p for p in perm_d])
    ncod, ndom, want = at.to_blocks(Tp, ncf, ndf)
    blocks, rows = ref.transform_tensor(data.block_inds, data.blocks, cod, dom, ncod, ndom, codomain_idcs, domain_idcs, mapping)
    assert len(want.blocks) > 0 and np.array_equal(rows, want.block_inds)
    for g, w in zip(blocks, want.blocks):
        assert np.array_equal(g, w)                            # a permutation: bit-exact
    res = ft.transform_tensor(_NpBackend(), data, cod, dom, ncod, ndom, codomain_idcs, domain_idcs, mapping)
    assert np.array_equal(res.block_inds, want.block_inds)
    for g, w in zip(res.blocks, want.blocks):
        assert np.array_equal(g, w)


def _random_data(rng, cod, dom, fill=0.7, cplx=False):
    rows, blocks = [], []
    for i, j in ft.common_sectors(cod, dom):
        if rng.random() < fill:
            sh = (cod.block_size(i), dom.block_size(j))
            blocks.append(rng.standard_normal(sh) + (1j * rng.standard_normal(sh) if cplx else 0))
            rows.append((i, j))
    return ft.FusionTreeData(rows, blocks)


def _dense(data, cod, dom):
    """block-diagonal dense matrix over (all codomain rows, all domain columns), sectors in their order"""
    ro = np.concatenate([[0], np.cumsum(cod.multiplicities)])
    co = np.concatenate([[0], np.cumsum(dom.multiplicities)])
    dtype = np.result_type(float, *[np.asarray(b).dtype for b in data.blocks])
    out = np.zeros((ro[-1], co[-1]), dtype=dtype)
    for (i, j), b in zip(data.block_inds.tolist(), data.blocks):
        out[ro[i]:ro[i + 1], co[j]:co[j + 1]] = b
    return out


def _spaces(rng):
    at = AbelianTrees(rng, J=2, K=2)
    cod, dom = at.space([0, 1]), at.space([2, 3])
    q = np.array([1.0, (1 + 5 ** 0.5) / 2, 2.0, 3.0, 1.5, 2.5, 1.0])
    cod.qdims, dom.qdims = q[:cod.num_sectors].copy(), q[:dom.num_sectors].copy()
    return cod, dom


def test_compose_and_decompositions_match_the_restatement_and_the_invariants(rng):
    bb = _NpBackend()
    cod, dom = _spaces(rng)
    mid = cod
    a = _random_data(rng, cod, mid)
    b = _random_data(rng, mid, dom)
    blocks, rows = ref.compose(a.block_inds, a.blocks, b.block_inds, b.blocks)
    c = ft.compose(bb, a, b)
    assert np.array_equal(c.block_inds, rows) and len(rows) > 0
    for x, y in zip(c.blocks, blocks):
        assert np.abs(x - y).max() <= 1e-13
    assert np.abs(_dense(c, cod, dom) - _dense(a, cod, mid) @ _dense(b, mid, dom)).max() <= 1e-12
    t = _random_data(rng, cod, dom, fill=0.6)
    new = None
    (ub, ui), (sb, si), (vb, vi) = ref.svd(t.block_inds, t.blocks, cod, dom)
    U, S, Vh = ft.svd(bb, t, cod, dom)
    assert np.array_equal(U.block_inds, ui) and np.array_equal(S.block_inds, si) and np.array_equal(Vh.block_inds, vi)
    for got, want in ((U.blocks, ub), (S.blocks, sb), (Vh.blocks, vb)):
        for x, y in zip(got, want):
            assert np.abs(np.asarray(x) - y).max(initial=0.0) <= 1e-13
    # U S Vh = t on the sectors with a block; U and Vh isometries on EVERY common sector (test_tensors.py:3393-3500)
    common = ft.common_sectors(cod, dom)
    s_of = {int(k): s for (k, _), s in zip(S.block_inds.tolist(), S.blocks)}
    t_of = {(int(i), int(j)): blk for (i, j), blk in zip(t.block_inds.tolist(), t.blocks)}
    for k, (i, j) in enumerate(common):
        u, vh = U.blocks[k], Vh.blocks[k]
        assert np.abs(u.T @ u - np.eye(u.shape[1])).max() <= 1e-13 and np.abs(vh @ vh.T - np.eye(vh.shape[0])).max() <= 1e-13
        if k in s_of:
            assert np.abs((u * s_of[k]) @ vh - t_of[(i, j)]).max() <= 1e-12
        else:
            assert (i, j) not in t_of
    for lq in (False, True):
        (qb, qi), (tb, ti) = ref.qr(t.block_inds, t.blocks, cod, dom, lq=lq)
        first, second = (ft.lq if lq else ft.qr)(bb, t, cod, dom)
        iso, tri = (second, first) if lq else (first, second)
        assert np.array_equal(iso.block_inds, qi) and np.array_equal(tri.block_inds, ti)
        for x, y in zip(iso.blocks, qb):
            assert np.abs(x - y).max() <= 1e-13
        for x, y in zip(tri.blocks, tb):
            assert np.abs(x - y).max() <= 1e-13
    h = ft.FusionTreeData(a.block_inds, [blk + blk.T for blk in a.blocks])
    w, v = ref.eigh(h.block_inds, h.blocks, cod)
    W, V = ft.eigh(bb, h, cod)
    assert len(V.blocks) == cod.num_sectors and len(W.blocks) == len(h.blocks)
    for x, y in zip(W.blocks, w):
        assert np.abs(x - y).max() <= 1e-13
    for x, y in zip(V.blocks, v):
        assert np.abs(np.abs(x) - np.abs(y)).max() <= 1e-10


def test_truncation_with_quantum_dimensions_matches_the_restatement(rng):
    bb = _NpBackend()
    cod, dom = _spaces(rng)
    for _ in range(20):
        rows, blocks = [], []
        for j in range(dom.num_sectors):
            if rng.random() < 0.7:
                rows.append((j, j))
                blocks.append(np.sort(rng.random(int(dom.multiplicities[j])))[::-1])
        if not rows:
            continue
        S = ft.FusionTreeData(rows, blocks)
        opts = dict(chi_max=int(rng.integers(1, int(dom.multiplicities.sum()) + 1)), trunc_cut=float(rng.choice([0.0, 0.2])))
        mb, mi, err, nn = ref.truncate_singular_values(S.block_inds, S.blocks, dom, **opts)
        gb, gi, gerr, gnn = ft.truncate_singular_values(bb, S, dom, **opts)
        assert np.array_equal(gi, mi) and abs(gerr - err) <= 1e-14 and abs(gnn - nn) <= 1e-14
        for x, y in zip(gb, mb):
            assert np.array_equal(x, y)
