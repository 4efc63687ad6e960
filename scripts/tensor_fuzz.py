"""Randomised soak of the tensor level (cyten_amd/abelian.py on the device backend): random symmetries (products of U(1) and Z_N),
random legs, sparse block tables -- compose against the dense contraction, combine_legs -> batched SVD -> dense reconstruction,
truncated SVD (eager and lazy) against the dense singular values, QR (thin / full) and eigh of the sector blocks,
linear_combination / inner / norm.
`python scripts/tensor_fuzz.py [n_rounds=100] [seed=0]`"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from cyten_amd import abelian as ab, workloads as wl
from cyten_amd.block_backend import HipBlockBackend

n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(seed)
bad = 0


def fail(tag, msg):
    global bad
    bad += 1
    print(f'[tensor-fuzz] FAIL {tag}: {msg}', flush=True)


def rand_leg(moduli, sign, max_mult):
    nsec = min(int(rng.integers(1, 6)), int(np.prod([m if m else 7 for m in moduli])))   # (no more sectors than the group has)
    secs = set()
    while len(secs) < nsec:
        secs.add(tuple(int(rng.integers(-3, 4)) if m == 0 else int(rng.integers(0, m)) for m in moduli))
    secs = sorted(secs)
    return wl.make_leg(moduli, np.array(secs), rng.integers(1, max_mult + 1, len(secs)), sign)


t0 = time.time()
for it in range(n_rounds):
    moduli = [(0,), (2,), (0, 2), (0, 0), (3,)][int(rng.integers(0, 5))]
    big = rng.random() < 0.3
    mm = 40 if big else 7
    vl, p1, mid, p2, vr = rand_leg(moduli, +1, mm), rand_leg(moduli, +1, 3), rand_leg(moduli, -1, mm), rand_leg(moduli, +1, 3), rand_leg(moduli, -1, mm)
    A = wl.random_tensor(moduli, [vl, p1, mid], rng, num_codomain=2, fill=float(rng.choice([1.0, 0.7])))
    B = wl.random_tensor(moduli, [wl.flip(mid), p2, vr], rng, num_codomain=1, fill=float(rng.choice([1.0, 0.7])))
    if rng.random() < 0.25:   # complex blocks
        for t in (A, B):
            t.blocks = [b + 1j * rng.standard_normal(b.shape) for b in t.blocks]
    a, b = ab.AbelianTensor.from_spec(bb, A), ab.AbelianTensor.from_spec(bb, B)
    da, db = a.to_dense(bb), b.to_dense(bb)
    theta = ab.compose(bb, a, b, 1)
    dense = np.tensordot(da, db, axes=([2], [0]))
    got = theta.to_dense(bb)
    scale = max(1.0, np.abs(dense).max())
    if got.shape != dense.shape or np.abs(got - dense).max() > 1e-11 * scale:
        fail('compose', f'round {it} moduli {moduli}')
        continue
    nrm = np.linalg.norm(dense)
    if abs(ab.norm(bb, theta) - nrm) > 1e-11 * max(1.0, nrm):
        fail('norm', f'round {it}')
    if len(theta.blocks) == 0:
        continue
    # tensor BLAS-1
    lc = ab.linear_combination(bb, 0.5, theta, -2.0, theta)
    if np.abs(lc.to_dense(bb) + 1.5 * dense).max() > 1e-11 * scale:
        fail('linear_combination', f'round {it}')
    ip = ab.inner(bb, theta, theta)
    if abs(ip - nrm ** 2) > 1e-10 * max(1.0, nrm ** 2):
        fail('inner', f'round {it}')
    # combine -> svd -> reconstruction of every sector; all singular values against the dense matrix
    mv = ab.combine_legs_to_matrix(bb, theta, 2)
    try:
        U, S, Vh = ab.svd(bb, mv)
    except Exception as e:   # keep the blocks for a reproduction
        import os
        os.makedirs('gpurun_out', exist_ok=True)
        np.savez(f'gpurun_out/tensor_fuzz_svd_fail_seed{seed}_round{it}.npz', *[bb.to_numpy(m) for m in mv.blocks])
        fail('svd raised', f'round {it} moduli {moduli}: {e}')
        continue
    s_all = np.sort(np.concatenate([bb.to_numpy(s) for s in S]))[::-1]
    dmat = dense.reshape(dense.shape[0] * dense.shape[1], -1)
    s_ref = np.linalg.svd(dmat, compute_uv=False)
    kk = min(len(s_all), len(s_ref))
    if np.abs(s_all[:kk] - s_ref[:kk]).max() > 1e-10 * max(1.0, s_ref[0]) or (len(s_ref) > kk and s_ref[kk:].max() > 1e-10 * max(1.0, s_ref[0])):
        fail('svd values', f'round {it} moduli {moduli}')
    for m, u, s, vh in zip(mv.blocks, U, S, Vh):
        m_, u_, s_, v_ = bb.to_numpy(m), bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
        k = len(s_)
        if np.abs((u_ * s_) @ v_ - m_).max() > 1e-10 * max(1.0, np.abs(m_).max()) or np.abs(u_.conj().T @ u_ - np.eye(k)).max() > 1e-10 or \
                np.abs(v_ @ v_.conj().T - np.eye(k)).max() > 1e-10:
            fail('svd factors', f'round {it} sector {m_.shape}')
    # QR (thin and full) and eigh (of m m^H) of the same sector blocks
    for full in (False, True):
        try:
            Q, R = ab.qr(bb, mv, full)
        except Exception as e:
            fail('qr raised', f'round {it} full {full}: {e}')
            continue
        for m, q, r in zip(mv.blocks, Q, R):
            m_, q_, r_ = bb.to_numpy(m), bb.to_numpy(q), bb.to_numpy(r)
            sc = max(1.0, np.abs(m_).max())
            if q_.shape[1] != r_.shape[0] or np.abs(q_ @ r_ - m_).max() > 1e-10 * sc or np.abs(q_.conj().T @ q_ - np.eye(q_.shape[1])).max() > 1e-10 \
                    or np.abs(np.tril(r_, -1)).max(initial=0.0) > 1e-10 * sc:
                fail('qr', f'round {it} full {full} sector {m_.shape}')
    hs = [bb.to_numpy(m) for m in mv.blocks]
    hs = [h @ h.conj().T for h in hs]
    try:
        res = bb.eigh_batched([bb.as_block(h) for h in hs])
        for h, (w, v) in zip(hs, res):
            w_, v_ = bb.to_numpy(w), bb.to_numpy(v)
            sc = max(1.0, np.abs(h).max())
            if np.abs(w_ - np.linalg.eigvalsh(h)).max() > 1e-10 * sc or np.abs((v_ * w_) @ v_.conj().T - h).max() > 1e-10 * sc or \
                    np.abs(v_.conj().T @ v_ - np.eye(len(w_))).max() > 1e-10:
                fail('eigh', f'round {it} sector {h.shape}')
    except Exception as e:
        fail('eigh raised', f'round {it}: {e}')
    # truncation, eager and lazy: kept values are the chi_max largest, err^2 the discarded weight
    chi = int(rng.integers(1, max(2, len(s_all))))
    for lazy in (False, True):
        mv2, Ut, St, Vt, err, new_norm = ab.truncated_svd(bb, theta, 2, lazy_null=lazy, chi_max=chi)
        kept = np.sort(np.concatenate([bb.to_numpy(s) for s in St]))[::-1]
        want = s_all[:min(chi, len(s_all))]
        want = want[want > 0]   # (tensor_backend.cpp:139-242: values whose cumulated weight is zero are never kept)
        if len(kept) != len(want) or np.abs(kept - want).max() > 1e-10 * max(1.0, s_ref[0]):
            # (ties at the cut may legitimately keep a different member of a degenerate group: compare the values only)
            fail('truncation kept', f'round {it} lazy {lazy} chi {chi}: {len(kept)} vs {len(want)}')
        disc = np.sum(s_all[len(want):] ** 2)
        if abs(err - disc) > 1e-9 * max(1.0, s_ref[0] ** 2) and abs(err - np.sqrt(disc)) > 1e-9 * max(1.0, s_ref[0]):
            fail('truncation err', f'round {it} lazy {lazy}: {err} vs {disc}')
        for m, u, s, vh in zip(mv2.blocks, Ut, St, Vt):
            u_, s_, v_ = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
            k = len(s_)
            if k and (np.abs(u_.conj().T @ u_ - np.eye(k)).max() > 1e-10 or np.abs(v_ @ v_.conj().T - np.eye(k)).max() > 1e-10):
                fail('truncated factors', f'round {it} lazy {lazy}')
    if it % 10 == 9:
        print(f'[tensor-fuzz] {it + 1} rounds, {bad} failures, {time.time() - t0:.0f} s', flush=True)
print(f'[tensor-fuzz] done: {n_rounds} rounds, seed {seed}: {bad} failures')
sys.exit(1 if bad else 0)
