"""Device side of the FusionTreeBackend callers (cyten_amd.fusion_tree) through the C-ABI:
`transform_tensor` -- the loops of TreePairMapping::transform_tensor (fusion_tree_mapping.cpp:391-513) emitting ONE
`cyb_lincomb_strided_batched_*` launch per tensor -- against the reference-held tree moves (Fibonacci C / B symbols, SU(3)_3 C
symbols: complex coefficients) and against dense leg permutations of abelian tensors with multiplicities > 1; compose / svd / qr /
lq / eigh of FusionTreeData (one grouped / batched call each, identity blocks for absent sectors, fusion_tree_backend.cpp:669-698,
:2033-2252) and the quantum-dimension weighted truncation (:2254-2340) against the oracle's per-block restatements."""
import numpy as np
import pytest

from cyten_amd import fusion_tree as ft
from fusion_tree_cases import AbelianTrees, spaces_from_fixture
from oracle import fusion_tree_ref as ref
from test_fusion_tree import _dense, _random_data, _spaces
from tree_move_fixture import expected, inputs, load

pytestmark = pytest.mark.gpu
CASES, SYM = load()


def _to_dev(bb, data):
    return ft.FusionTreeData(data.block_inds, [bb.as_block(b) for b in data.blocks])


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_device_transform_tensor_reproduces_the_reference_held_tree_moves(bb, case, rng):
    old = inputs(case, rng)
    want, mask = expected(case, SYM, old)
    cod, dom, ncod, ndom, mapping = spaces_from_fixture(case, SYM)
    lg = case['legs']
    data = ft.FusionTreeData([(b, b) for b in range(len(old))], [bb.as_block(o) for o in old])
    res = ft.transform_tensor(bb, data, cod, dom, ncod, ndom, lg['codomain_idcs'], lg['domain_idcs'], mapping)
    res = ft.discard_zero_blocks(bb, res, 1e-14)
    got = {tuple(r): bb.to_numpy(b) for r, b in zip(res.block_inds.tolist(), res.blocks)}
    for nb, (w, m) in enumerate(zip(want, mask)):
        if not m.any():
            assert (nb, nb) not in got
            continue
        g = got[(nb, nb)]
        assert g.dtype == np.complex128 and np.abs(g - w)[m].max() <= 1e-14 and np.abs(g[~m]).max(initial=0.0) == 0.0


@pytest.mark.parametrize('perm_c,perm_d', [((1, 0, 2), (0, 1)), ((0, 2, 1), (1, 0)), ((2, 0, 1), (1, 0))])
@pytest.mark.parametrize('cplx', [False, True])
def test_device_transform_tensor_is_the_dense_leg_permutation_for_abelian_trees(bb, rng, perm_c, perm_d, cplx):
    at = AbelianTrees(rng)
    T = at.dense(rng, cplx)
    cod, dom, data = at.to_blocks(T, range(at.J), range(at.J, at.J + at.K))
    codomain_idcs, domain_idcs, ncf, ndf, mapping = at.braid(perm_c, perm_d)
    ncod, ndom, want = at.to_blocks(np.transpose(T, list(perm_c) + [at.J + p for p in perm_d]), ncf, ndf)
    res = ft.transform_tensor(bb, _to_dev(bb, data), cod, dom, ncod, ndom, codomain_idcs, domain_idcs, mapping)
    assert np.array_equal(res.block_inds, want.block_inds)
    for g, w in zip(res.blocks, want.blocks):
        assert np.array_equal(bb.to_numpy(g), w)               # a permutation with coefficient 1: bit-exact


def test_device_compose_and_decompositions(bb, rng):
    cod, dom = _spaces(rng)
    a = _random_data(rng, cod, cod)
    b = _random_data(rng, cod, dom)
    blocks, rows = ref.compose(a.block_inds, a.blocks, b.block_inds, b.blocks)
    c = ft.compose(bb, _to_dev(bb, a), _to_dev(bb, b))
    assert np.array_equal(c.block_inds, rows)
    for x, y in zip(c.blocks, blocks):
        assert np.abs(bb.to_numpy(x) - y).max() <= 1e-10 * max(1.0, np.abs(y).max())
    t = _random_data(rng, cod, dom, fill=0.6)
    td = _to_dev(bb, t)
    (ub, ui), (sb, si), (vb, vi) = ref.svd(t.block_inds, t.blocks, cod, dom)
    U, S, Vh = ft.svd(bb, td, cod, dom)
    assert np.array_equal(U.block_inds, ui) and np.array_equal(S.block_inds, si) and np.array_equal(Vh.block_inds, vi)
    for x, y in zip(S.blocks, sb):
        assert np.abs(bb.to_numpy(x) - y).max(initial=0.0) <= 1e-10 * max(1.0, np.abs(y).max(initial=0.0))
    s_of = {int(k): bb.to_numpy(s) for (k, _), s in zip(S.block_inds.tolist(), S.blocks)}
    t_of = {(int(i), int(j)): blk for (i, j), blk in zip(t.block_inds.tolist(), t.blocks)}
    for k, (i, j) in enumerate(ft.common_sectors(cod, dom)):
        u, vh = bb.to_numpy(U.blocks[k]), bb.to_numpy(Vh.blocks[k])
        assert np.abs(u.T @ u - np.eye(u.shape[1])).max() <= 1e-10 and np.abs(vh @ vh.T - np.eye(vh.shape[0])).max() <= 1e-10
        if k in s_of:
            assert np.abs((u * s_of[k]) @ vh - t_of[(i, j)]).max() <= 1e-10 * max(1.0, np.abs(t_of[(i, j)]).max())
    for lq in (False, True):
        (qb, qi), (tb, ti) = ref.qr(t.block_inds, t.blocks, cod, dom, lq=lq)
        first, second = (ft.lq if lq else ft.qr)(bb, td, cod, dom)
        iso, tri = (second, first) if lq else (first, second)
        assert np.array_equal(iso.block_inds, qi) and np.array_equal(tri.block_inds, ti)
        for x, y in zip(iso.blocks, qb):
            assert np.abs(bb.to_numpy(x) - y).max() <= 1e-10
        for x, y in zip(tri.blocks, tb):
            assert np.abs(bb.to_numpy(x) - y).max() <= 1e-10 * max(1.0, np.abs(y).max())
    h = ft.FusionTreeData(a.block_inds, [blk + blk.T for blk in a.blocks])
    w, v = ref.eigh(h.block_inds, h.blocks, cod)
    W, V = ft.eigh(bb, _to_dev(bb, h), cod)
    assert len(V.blocks) == cod.num_sectors
    for x, y in zip(W.blocks, w):
        assert np.abs(bb.to_numpy(x) - y).max() <= 1e-10 * max(1.0, np.abs(y).max())
    assert abs(ft.norm(bb, td, cod) - np.sqrt(sum(cod.qdims[i] * np.linalg.norm(blk) ** 2 for (i, _), blk in zip(t.block_inds.tolist(), t.blocks)))) <= 1e-10


def test_device_truncation_with_quantum_dimensions(bb, rng):
    cod, dom = _spaces(rng)
    for _ in range(15):
        rows, blocks = [], []
        for j in range(dom.num_sectors):
            if rng.random() < 0.7:
                rows.append((j, j))
                blocks.append(np.sort(rng.random(int(dom.multiplicities[j])))[::-1])
        if not rows:
            continue
        opts = dict(chi_max=int(rng.integers(1, int(dom.multiplicities.sum()) + 1)), trunc_cut=float(rng.choice([0.0, 0.2])))
        mb, mi, err, nn = ref.truncate_singular_values(np.array(rows), blocks, dom, **opts)
        gb, gi, gerr, gnn = ft.truncate_singular_values(bb, ft.FusionTreeData(rows, [bb.as_block(x) for x in blocks]), dom, **opts)
        assert np.array_equal(gi, mi) and abs(gerr - err) <= 1e-12 and abs(gnn - nn) <= 1e-12
        for x, y in zip(gb, mb):
            assert np.array_equal(x, y)
