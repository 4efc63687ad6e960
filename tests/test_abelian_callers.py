"""The remaining AbelianBackend callers of SURVEY.md section 8 row a10 on the CPU: the oracle's restatements
(oracle/abelian_ref.py: partial_compose abelian.cpp:2853-2951, _mask_contract :2484-2583, qr :3084-3151, lq :2304-2385)
against the dense contractions they stand for -- the reference's own criterion (test_tensors.py compares with
``to_numpy()`` results) -- and the host logic of cyten_amd.abelian on the numpy stand-in against the oracle, bit-identical
block tables."""
import numpy as np
import pytest

from abelian_caller_cases import dense_partial_compose, partial_compose_cases, two_leg_cases
from cyten_amd import abelian as ab
from cyten_amd import workloads as wl
from numpy_backend import NumpyGroupedBackend
from oracle import abelian_ref as ref


def _dense(data):
    return ref.to_dense(data)


@pytest.mark.parametrize('case', range(9))
def test_partial_compose_oracle_and_host_logic(case):
    a, b, first = partial_compose_cases()[case]
    got = ref.partial_compose(a, b, first)
    want = dense_partial_compose(ref.to_dense(a), ref.to_dense(b), a, b, first)
    assert np.abs(_dense(got) - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    assert np.array_equal(got.block_inds, got.block_inds[np.lexsort(got.block_inds.T)])
    bb = NumpyGroupedBackend()
    res = ab.partial_compose(bb, ab.AbelianTensor.from_spec(bb, a), ab.AbelianTensor.from_spec(bb, b), first)
    assert np.array_equal(res.block_inds, got.block_inds) and res.num_codomain == got.num_codomain
    for x, y in zip(res.blocks, got.blocks):
        assert x.shape == y.shape and np.abs(x - y).max() <= 1e-12 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize('leg_idx', [0, 1, 2, 3])
def test_mask_contract_oracle_and_host_logic(leg_idx, rng):
    a = partial_compose_cases()[0][0]
    t = ab.AbelianTensor.from_spec(NumpyGroupedBackend(), a)
    leg = t.legs[leg_idx]
    flags = rng.random(leg.dim) < 0.6
    drop = int(np.unique(a.block_inds[:, leg_idx])[-1])             # a sector that HAS blocks loses every state: they are dropped
    flags[int(leg.slices[drop]):int(leg.slices[drop + 1])] = False
    flags[int(leg.slices[drop - 1 if drop else 1])] = True           # (something survives, whatever the draw)
    mask = ab.Mask.from_flags(leg, flags)
    small = wl.LegSpec(mask.small_leg.sectors, mask.small_leg.mults, leg.sign)
    got = ref.mask_contract(a, mask.blocks, mask.block_inds, leg_idx, True, small)
    dense = np.compress(flags, ref.to_dense(a), axis=leg_idx)
    assert np.abs(_dense(got) - dense).max() == 0.0
    bb = NumpyGroupedBackend()
    res = ab.mask_contract(bb, t, mask, leg_idx, True)
    assert np.array_equal(res.block_inds, got.block_inds)
    for x, y in zip(res.blocks, got.blocks):
        assert np.array_equal(x, y)
    # ... and back: embedding the projected tensor gives the original with the masked-out states zeroed
    back = ref.mask_contract(got, mask.blocks, mask.block_inds, leg_idx, False, a.legs[leg_idx])
    want = ref.to_dense(a) * flags.reshape([-1 if k == leg_idx else 1 for k in range(len(a.legs))])
    assert np.abs(_dense(back) - want).max() == 0.0
    res2 = ab.mask_contract(bb, res, mask, leg_idx, False)
    assert np.array_equal(res2.block_inds, back.block_inds)
    for x, y in zip(res2.blocks, back.blocks):
        assert np.array_equal(x, y)


@pytest.mark.parametrize('case', range(3))
@pytest.mark.parametrize('lq', [False, True])
def test_two_leg_qr_lq_oracle_and_host_logic(case, lq):
    t = two_leg_cases()[case]
    (b0, r0), (b1, r1), common = (ref.lq_two_leg if lq else ref.qr_two_leg)(t)
    cod, dom = t.legs
    new_mults = [min(int(cod.mults[j]), int(dom.mults[k])) for j, k in common]
    new_in = wl.make_leg((0,), cod.sectors[[j for j, _ in common]], new_mults, +1)
    new_out = wl.flip(new_in)
    first = ref._Data(t.moduli, [cod, new_out], r0, b0, 1)
    second = ref._Data(t.moduli, [new_in, dom], r1, b1, 1)
    want = ref.to_dense(t)
    assert np.abs(ref.to_dense(first) @ ref.to_dense(second) - want).max() <= 1e-13 * max(1.0, np.abs(want).max())
    iso = ref.to_dense(second) if lq else ref.to_dense(first)
    gram = iso @ iso.T if lq else iso.T @ iso                          # an isometry on EVERY common sector, also where t has no block
    assert np.abs(gram - np.eye(gram.shape[0])).max() <= 1e-13
    bb = NumpyGroupedBackend()
    f, s = (ab.lq_tensor if lq else ab.qr_tensor)(bb, ab.AbelianTensor.from_spec(bb, t))
    o0 = np.lexsort(np.asarray(r0).T) if len(r0) else []
    o1 = np.lexsort(np.asarray(r1).T) if len(r1) else []
    assert np.array_equal(f.block_inds, np.asarray(r0)[o0].reshape(-1, 2)) and np.array_equal(s.block_inds, np.asarray(r1)[o1].reshape(-1, 2))
    for x, i in zip(f.blocks, o0):
        assert np.abs(x - b0[i]).max(initial=0.0) <= 1e-13
    for x, i in zip(s.blocks, o1):
        assert np.abs(x - b1[i]).max(initial=0.0) <= 1e-13


def test_to_block_backend_and_move_to_device_keep_the_tensor():
    t = two_leg_cases()[0]
    bb = NumpyGroupedBackend()
    x = ab.AbelianTensor.from_spec(bb, t)
    y = ab.to_block_backend(bb, x, bb_old=bb)
    z = ab.move_to_device(bb, y, 'cpu')
    assert np.array_equal(z.block_inds, x.block_inds)
    for p, q in zip(z.blocks, x.blocks):
        assert np.array_equal(p, q)


def test_weighted_truncation_selection_host_mirror_matches_the_oracle(rng):
    """qdims-weighted selection (tensor_backend.cpp:158-164): host mirror = oracle restatement"""
    for _ in range(50):
        sizes = rng.integers(1, 9, rng.integers(1, 6))
        S = np.concatenate([np.sort(rng.random(n))[::-1] for n in sizes])
        q = np.concatenate([np.full(n, rng.choice([1.0, (1 + 5 ** 0.5) / 2, 2.0, 3.0])) for n in sizes])
        opts = dict(chi_max=int(rng.integers(1, len(S) + 1)), trunc_cut=float(rng.choice([0.0, 0.1, 0.5])),
                    degeneracy_tol=float(rng.choice([0.0, 1e-2])))
        m0, e0, n0 = ref.truncation_selection(S, qdims=q, **opts)
        m1, e1, n1 = ab.truncation_selection(S, qdims=q, **opts)
        assert np.array_equal(m0, m1) and e0 == e1 and n0 == n1
