"""Lanczos matvec group: H_eff matvec and the Lanczos ground state.

CPU part: the oracle restatement of the reference's Lanczos is pinned against numpy.linalg.eigh
(known answers), and the product's HOST logic (permute_legs, plan reuse, tensor axpy, the Lanczos
driver) runs on the numpy stand-in backend against the dense oracle.  GPU part: the same checks through
the C-ABI on the device (fp64, tolerances written in the tests)."""
import numpy as np
import pytest

from cyten_amd import abelian as ab
from cyten_amd import krylov
from cyten_amd import workloads as wl
from oracle import abelian_ref as ref
from oracle import krylov_ref

from helpers import to_device_tensor
from numpy_backend import NumpyGroupedBackend


def _dense(spec):
    return ref.to_dense(spec)


def _setup(bbk, chi, D, charged, seed=7):
    cfg = wl.config_heff(chi, D, seed=seed, charged_mpo=charged)
    dev = {k: to_device_tensor(bbk, v) for k, v in cfg.items()}
    dense = {k: _dense(v) for k, v in cfg.items()}
    H = krylov.HEffective(bbk, dev['LP'], dev['W1'], dev['W2'], dev['RP'])
    return cfg, dev, dense, H


# ------------------------------------------------------------------------------------------ oracle pin

@pytest.mark.parametrize('n,seed', [(12, 0), (60, 1), (200, 2)])
def test_oracle_lanczos_against_eigh(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n))
    A = A + A.T
    w, V = np.linalg.eigh(A)
    E0, psi, N = krylov_ref.lanczos_dense(lambda v: A @ v, rng.standard_normal(n), N_max=n, reortho=True)
    assert abs(E0 - w[0]) < 1e-10 * max(1.0, abs(w[0]))
    assert abs(abs(psi @ V[:, 0]) - 1.0) < 1e-8
    assert N <= n


def test_oracle_lanczos_defaults_stop_early():
    """With the reference's defaults (N_max=20, P_tol=1e-14) the iteration stops at N_max on a generic
    matrix and still returns a normalised Ritz vector whose energy is the smallest Ritz value."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((300, 300))
    A = A + A.T
    E0, psi, N = krylov_ref.lanczos_dense(lambda v: A @ v, rng.standard_normal(300))
    assert N == 20
    assert abs(np.linalg.norm(psi) - 1.0) < 1e-12
    assert abs(psi @ A @ psi - E0) < 1e-8 * abs(E0)


# ------------------------------------------------------------------------------------------ host logic (CPU)

@pytest.mark.parametrize('charged', [False, True])
def test_heff_matvec_host_logic(charged):
    nbk = NumpyGroupedBackend()
    cfg, dev, dense, H = _setup(nbk, 40, 3, charged)
    out = H.matvec(dev['theta'])
    expect = krylov_ref.heff_dense(dense['LP'], dense['W1'], dense['W2'], dense['RP'])(dense['theta'])
    got = out.to_dense(nbk)
    assert got.shape == expect.shape
    np.testing.assert_allclose(got, expect, rtol=0, atol=1e-10 * np.abs(expect).max())
    # second call reuses the cached sector matching and gives the same blocks
    out2 = H.matvec(dev['theta'])
    for x, y in zip(out.blocks, out2.blocks):
        np.testing.assert_array_equal(x, y)
    assert H.flops_per_matvec > 0


def test_linear_combination_union_of_blocks():
    nbk = NumpyGroupedBackend()
    cfg = wl.config_heff(24, 2, seed=1)
    t = cfg['theta']
    rng = np.random.default_rng(0)
    keep_a = rng.random(len(t.blocks)) < 0.7
    keep_b = rng.random(len(t.blocks)) < 0.7
    def sub(keep):
        return wl.TensorSpec(t.moduli, t.legs, t.block_inds[keep], [b for b, k in zip(t.blocks, keep) if k], t.num_codomain)
    A, B = sub(keep_a), sub(keep_b)
    a, b = to_device_tensor(nbk, A), to_device_tensor(nbk, B)
    c = ab.linear_combination(nbk, 2.0, a, -0.5, b)
    np.testing.assert_allclose(c.to_dense(nbk), 2.0 * _dense(A) - 0.5 * _dense(B), atol=1e-13)
    assert len(c.blocks) == int(np.sum(keep_a | keep_b))
    assert abs(ab.inner(nbk, a, b) - np.sum(_dense(A) * _dense(B))) < 1e-10


def test_lanczos_host_logic_matches_oracle():
    nbk = NumpyGroupedBackend()
    cfg, dev, dense, H = _setup(nbk, 24, 3, False)
    mv = krylov_ref.heff_dense(dense['LP'], dense['W1'], dense['W2'], dense['RP'])
    opts = dict(N_max=30, reortho=True)
    E0, psi, N = krylov.lanczos(nbk, H, dev['theta'], opts)
    E0r, psir, Nr = krylov_ref.lanczos_dense(mv, dense['theta'], **opts)
    assert N == Nr
    assert abs(E0 - E0r) < 1e-9 * abs(E0r)
    assert abs(abs(np.sum(psi.to_dense(nbk) * psir)) - 1.0) < 1e-8
    # the dense spectrum restricted to theta's charge sector agrees (H_eff is Hermitian by construction)
    Hm = krylov_ref.heff_matrix(dense['LP'], dense['W1'], dense['W2'], dense['RP'])
    assert np.abs(Hm - Hm.T).max() < 1e-12 * np.abs(Hm).max()


def test_lanczos_small_cache_rebuild():
    """N_cache < N: the dropped Krylov vectors are regenerated (krylov_based.cpp:896-920)."""
    nbk = NumpyGroupedBackend()
    cfg, dev, dense, H = _setup(nbk, 24, 3, False)
    E_full, psi_full, N_full = krylov.lanczos(nbk, H, dev['theta'], dict(N_max=12))
    E_small, psi_small, N_small = krylov.lanczos(nbk, H, dev['theta'], dict(N_max=12, N_cache=3))
    assert N_full == N_small
    assert abs(E_full - E_small) < 1e-10 * abs(E_full)
    assert abs(abs(ab.inner(nbk, psi_full, psi_small)) - 1.0) < 1e-8


# ------------------------------------------------------------------------------------------ device (GPU)

@pytest.mark.gpu
@pytest.mark.parametrize('chi,D,charged', [(48, 3, False), (96, 5, True), (160, 5, False)])
def test_gpu_heff_matvec(bb, chi, D, charged):
    cfg, dev, dense, H = _setup(bb, chi, D, charged)
    out = H.matvec(dev['theta'])
    expect = krylov_ref.heff_dense(dense['LP'], dense['W1'], dense['W2'], dense['RP'])(dense['theta'])
    np.testing.assert_allclose(out.to_dense(bb), expect, rtol=0, atol=1e-10 * np.abs(expect).max())


@pytest.mark.gpu
def test_gpu_lanczos_ground_state(bb):
    cfg, dev, dense, H = _setup(bb, 64, 4, False)
    mv = krylov_ref.heff_dense(dense['LP'], dense['W1'], dense['W2'], dense['RP'])
    opts = dict(N_max=40, reortho=True)
    E0, psi, N = krylov.lanczos(bb, H, dev['theta'], opts)
    E0r, psir, Nr = krylov_ref.lanczos_dense(mv, dense['theta'], **opts)
    assert abs(N - Nr) <= 1
    assert abs(E0 - E0r) < 1e-9 * abs(E0r)
    assert abs(abs(np.sum(psi.to_dense(bb) * psir)) - 1.0) < 1e-7
    assert abs(ab.norm(bb, psi) - 1.0) < 1e-12
    # the flat pools span every charge-allowed block, the RESULT only the blocks some Krylov vector holds (the start vector's
    # and what the operator produced), and the solver object can run again from the caller's tensor (round-2 advisor findings)
    solver = krylov.LanczosGroundState(bb, H, dev['theta'], opts)
    E1, psi1, _ = solver.run()
    assert solver.psi0 is dev['theta']
    E2, psi2, _ = solver.run()
    assert abs(E1 - E0) < 1e-10 * abs(E0) and abs(E2 - E0) < 1e-10 * abs(E0)
    allowed = ab.AbelianTensor.allowed_block_inds(dev['theta'].symmetry, dev['theta'].legs)
    assert len(psi1.blocks) <= len(allowed)
    start = dev['theta']
    few = ab.AbelianTensor(start.symmetry, start.legs, start.blocks[:1], start.block_inds[:1], start.num_codomain)   # one-block start vector
    E3, psi3, _ = krylov.lanczos(bb, H, few, dict(N_max=3))
    norms = [bb.norm(b) for b in psi3.blocks]
    assert len(psi3.blocks) <= len(allowed) and all(nrm > 0 for nrm in norms)        # no explicit all-zero blocks in the result


def _complex_hermitian_heff(rng, chi=48, D=3, seed=5):
    """config_heff with complex Hermitian environments: LP[a,l,a'] = conj(LP[a',l,a]), RP[r,b',b] = conj(RP[r,b,b'])
    block by block (the MPO bond is uncharged, so every block maps onto itself)."""
    cfg = wl.config_heff(chi, D, seed=seed, charged_mpo=False)
    for i, blk in enumerate(cfg['LP'].blocks):
        z = blk + 1j * rng.standard_normal(blk.shape)
        cfg['LP'].blocks[i] = 0.5 * (z + np.conj(z.transpose(2, 1, 0)))
    for i, blk in enumerate(cfg['RP'].blocks):
        z = blk + 1j * rng.standard_normal(blk.shape)
        cfg['RP'].blocks[i] = 0.5 * (z + np.conj(z.transpose(0, 2, 1)))
    cfg['theta'].blocks = [b + 1j * rng.standard_normal(b.shape) for b in cfg['theta'].blocks]
    return cfg


def test_oracle_lanczos_complex_hermitian_against_eigh(rng):
    """The oracle's Lanczos on a complex Hermitian H_eff (alpha = Re<w|v>, krylov_based.cpp:861) finds numpy.linalg.eigh's
    lowest eigenvalue of the dense matrix."""
    cfg = _complex_hermitian_heff(rng, chi=16, D=2)
    dense = {k: _dense(v) for k, v in cfg.items()}
    Hm = krylov_ref.heff_matrix(dense['LP'], dense['W1'], dense['W2'], dense['RP'])
    np.testing.assert_allclose(Hm, Hm.conj().T, atol=1e-12 * np.abs(Hm).max())
    mv = krylov_ref.heff_dense(dense['LP'], dense['W1'], dense['W2'], dense['RP'])
    E0, psi, N = krylov_ref.lanczos_dense(mv, dense['theta'], N_max=80, reortho=True, P_tol=1e-22)
    w = np.linalg.eigvalsh(Hm)
    # theta lives in the charge-0 sector only: the Lanczos value is an eigenvalue of H restricted to it
    assert np.abs(w - E0).min() < 1e-8 * np.abs(w).max()
    assert abs(np.vdot(psi, mv(psi)).real - E0) < 1e-8 * np.abs(w).max() and abs(np.linalg.norm(psi) - 1) < 1e-12


@pytest.mark.gpu
def test_gpu_lanczos_complex_hermitian(bb, rng):
    """LanczosGroundState must consume the complex H_eff matvec (alpha is the REAL part of the overlap)."""
    cfg = _complex_hermitian_heff(rng)
    dev = {k: to_device_tensor(bb, v) for k, v in cfg.items()}
    dense = {k: _dense(v) for k, v in cfg.items()}
    H = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'])
    mv = krylov_ref.heff_dense(dense['LP'], dense['W1'], dense['W2'], dense['RP'])
    for opts in (dict(N_max=30, reortho=True), dict(N_max=12, N_cache=4)):
        E0, psi, N = krylov.lanczos(bb, H, dev['theta'], opts)
        E0r, psir, Nr = krylov_ref.lanczos_dense(mv, dense['theta'], **{k: v for k, v in opts.items() if k != 'N_cache'})
        assert abs(N - Nr) <= 1
        assert abs(E0 - E0r) < 1e-9 * abs(E0r)
        assert abs(abs(np.vdot(psir, psi.to_dense(bb))) - 1.0) < 1e-7
        assert abs(ab.norm(bb, psi) - 1.0) < 1e-12
    x, y = dev['theta'].blocks[0], dev['theta'].blocks[1] if len(dev['theta'].blocks) > 1 else dev['theta'].blocks[0]
    # inner(a, b, do_dagger=False) does NOT conjugate (numpy.cpp:816-842); complex operands
    from oracle import block_ops as ops
    a_np, b_np = bb.to_numpy(x), bb.to_numpy(x).transpose(3, 2, 1, 0).copy() * (0.3 - 0.7j)
    got = bb.inner(x, bb.as_block(b_np), False)
    assert abs(got - ops.inner(a_np, b_np, False)) <= 1e-11 * np.linalg.norm(a_np) * np.linalg.norm(b_np)
    assert abs(bb.inner(x, x, True) - ops.inner(a_np, a_np, True)) <= 1e-11 * np.linalg.norm(a_np) ** 2


@pytest.mark.gpu
@pytest.mark.parametrize('charged', [False, True])
def test_gpu_heff_replay_is_bit_identical(bb, charged):
    """The recorded launch sequence (cyten_amd/replay.py) re-issued on other vectors of the same block layout gives
    exactly the blocks of the ordinary path; a vector with another layout (blocks in separate buffers vs carved out of
    one pool, or missing blocks) gets a recording of its own."""
    cfg, dev, dense, H = _setup(bb, 96, 5, charged)
    plain = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'], replay=False)
    th = dev['theta']
    first = H.matvec(th)                                 # recorded
    again = H.matvec(th)                                 # replayed on the same buffers
    assert H.n_replayed == 1
    ref = plain.matvec(th)
    for x, y, z in zip(first.blocks, again.blocks, ref.blocks):
        np.testing.assert_array_equal(bb.to_numpy(x), bb.to_numpy(z))
        np.testing.assert_array_equal(bb.to_numpy(y), bb.to_numpy(z))
    # other vectors: pooled layout (what Lanczos produces), two different ones -> one recording, one replay
    same = [tuple(r) for r in th.block_inds] == [tuple(r) for r in ref.block_inds]
    v1 = ab.linear_combination(bb, 0.3, th, -1.7, ref) if same else ab.scale(bb, 0.3, th)
    v2 = ab.scale(bb, -2.5, v1)
    n0 = H.n_replayed
    o1, o2 = H.matvec(v1), H.matvec(v2)
    assert H.n_replayed == n0 + 1
    for v, o in ((v1, o1), (v2, o2)):
        r = plain.matvec(v)
        np.testing.assert_array_equal(o.block_inds, r.block_inds)
        for x, z in zip(o.blocks, r.blocks):
            np.testing.assert_array_equal(bb.to_numpy(x), bb.to_numpy(z))
    # linearity across replays (the replayed result buffers are fresh each time, nothing aliases)
    s = ab.linear_combination(bb, 1.0, o1, 1.0, o2)
    r = plain.matvec(ab.linear_combination(bb, 1.0, v1, 1.0, v2))
    np.testing.assert_allclose(s.to_dense(bb), r.to_dense(bb), rtol=0, atol=1e-10 * np.abs(r.to_dense(bb)).max())


@pytest.mark.gpu
@pytest.mark.parametrize('charged', [False, True])
def test_gpu_heff_shared_cache_relocates_operator(bb, charged):
    """One recording cache for several operators (what a DMRG run keeps across bonds and sweeps): a second operator
    with the SAME block layouts but other buffers and other values is served by replays only and gives exactly the
    blocks of the ordinary path; an operator with another layout records for itself."""
    cache = {}
    cfg, dev, dense, _ = _setup(bb, 96, 5, charged, seed=7)
    H1 = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'], cache=cache)
    H1.matvec(dev['theta'])
    assert (H1.n_recorded, H1.n_replayed) == (1, 0)
    # same structure, fresh buffers, different numbers (the same workload drawn with another seed has the same legs)
    cfg2, dev2, dense2, _ = _setup(bb, 96, 5, charged, seed=8)
    same_layout = all(np.array_equal(cfg[k].block_inds, cfg2[k].block_inds) for k in cfg)
    H2 = krylov.HEffective(bb, dev2['LP'], dev2['W1'], dev2['W2'], dev2['RP'], cache=cache)
    plain2 = krylov.HEffective(bb, dev2['LP'], dev2['W1'], dev2['W2'], dev2['RP'], replay=False)
    out = H2.matvec(dev2['theta'])
    ref2 = plain2.matvec(dev2['theta'])
    if same_layout:
        assert (H2.n_recorded, H2.n_replayed) == (0, 1)
        assert H2.flops_per_matvec == plain2.flops_per_matvec
    np.testing.assert_array_equal(out.block_inds, ref2.block_inds)
    for x, z in zip(out.blocks, ref2.blocks):
        np.testing.assert_array_equal(bb.to_numpy(x), bb.to_numpy(z))
    expect = krylov_ref.heff_dense(dense2['LP'], dense2['W1'], dense2['W2'], dense2['RP'])(dense2['theta'])
    np.testing.assert_allclose(out.to_dense(bb), expect, rtol=0, atol=1e-10 * np.abs(expect).max())
    # the first operator is still served correctly from the shared cache
    again = H1.matvec(dev['theta'])
    expect1 = krylov_ref.heff_dense(dense['LP'], dense['W1'], dense['W2'], dense['RP'])(dense['theta'])
    np.testing.assert_allclose(again.to_dense(bb), expect1, rtol=0, atol=1e-10 * np.abs(expect1).max())
    # another layout (smaller bond dimension) -> its own recording
    cfg3, dev3, dense3, _ = _setup(bb, 64, 5, charged, seed=9)
    H3 = krylov.HEffective(bb, dev3['LP'], dev3['W1'], dev3['W2'], dev3['RP'], cache=cache)
    out3 = H3.matvec(dev3['theta'])
    assert H3.n_recorded == 1
    expect3 = krylov_ref.heff_dense(dense3['LP'], dense3['W1'], dense3['W2'], dense3['RP'])(dense3['theta'])
    np.testing.assert_allclose(out3.to_dense(bb), expect3, rtol=0, atol=1e-10 * np.abs(expect3).max())


@pytest.mark.gpu
@pytest.mark.parametrize('complex_operator', [False, True])
def test_gpu_heff_replay_complex(bb, rng, complex_operator):
    """complex128 vectors (and operators): the launch sequence of a complex matvec -- operand expansion, real grouped
    GEMM on the interleaved storage, 16-byte leg rotations -- is recorded and replayed like the real one, bit-identical
    to the ordinary path and equal to the dense contraction."""
    cfg = wl.config_heff(64, 4, seed=11, charged_mpo=True)
    for key in (['theta'] + (['LP', 'RP'] if complex_operator else [])):
        cfg[key].blocks = [b + 1j * rng.standard_normal(b.shape) for b in cfg[key].blocks]
    dev = {k: to_device_tensor(bb, v) for k, v in cfg.items()}
    dense = {k: _dense(v) for k, v in cfg.items()}
    H = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'])
    plain = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'], replay=False)
    th = dev['theta']
    first, again, ref_out = H.matvec(th), H.matvec(th), plain.matvec(th)
    assert (H.n_recorded, H.n_replayed) == (1, 1)
    for x, y, z in zip(first.blocks, again.blocks, ref_out.blocks):
        assert x.is_complex
        np.testing.assert_array_equal(bb.to_numpy(x), bb.to_numpy(z))
        np.testing.assert_array_equal(bb.to_numpy(y), bb.to_numpy(z))
    v = ab.scale(bb, 0.5, ab.linear_combination(bb, 1.0, th, -2.0, ref_out)) if np.array_equal(th.block_inds, ref_out.block_inds) else ab.scale(bb, 0.5, th)
    o1, o2 = H.matvec(v), H.matvec(ab.scale(bb, 3.0, v))
    r1 = plain.matvec(v)
    for x, z in zip(o1.blocks, r1.blocks):
        np.testing.assert_array_equal(bb.to_numpy(x), bb.to_numpy(z))
    np.testing.assert_allclose(o2.to_dense(bb), 3.0 * r1.to_dense(bb), rtol=0, atol=1e-10 * np.abs(r1.to_dense(bb)).max())
    expect = krylov_ref.heff_dense(dense['LP'], dense['W1'], dense['W2'], dense['RP'])(dense['theta'])
    np.testing.assert_allclose(first.to_dense(bb), expect, rtol=0, atol=1e-10 * np.abs(expect).max())
