"""Randomised soak of the batched SVD (both forms) / QR / eigh (real and complex): lists of random shapes, ranks, gradings and scales against
numpy's singular values and the reference tests' invariants.  `python scripts/svd_fuzz.py [n_lists=40] [seed=0]`"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from cyten_amd.block_backend import HipBlockBackend

n_lists = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(seed)


def rand_matrix(cplx):
    m, n = (int(x) for x in rng.integers(1, 700, 2))
    if rng.random() < 0.15:
        m, n = int(rng.integers(1500, 2100)), int(rng.integers(40, 300))
        if rng.random() < 0.5:
            m, n = n, m
    kind = rng.integers(0, 6)
    g = (lambda s: rng.standard_normal(s) + 1j * rng.standard_normal(s)) if cplx else rng.standard_normal
    k = min(m, n)
    if kind == 0:
        a = g((m, n))
    elif kind == 1:                                  # theta-like: about half rank
        r = max(1, k // 2 + int(rng.integers(-3, 4)))
        a = g((m, r)) @ g((r, n))
    elif kind == 2:                                  # very low rank
        r = int(rng.integers(1, 6))
        a = g((m, r)) @ g((r, n))
    elif kind == 3:                                  # graded columns
        a = g((m, n)) * np.logspace(0, -int(rng.integers(4, 15)), n)
    elif kind == 4:                                  # low rank + noise at a random level
        r = max(1, k // 3)
        a = g((m, r)) @ g((r, n)) + 10.0 ** (-int(rng.integers(6, 15))) * g((m, n))
    else:                                            # clustered values
        q1, _ = np.linalg.qr(g((m, k)))
        q2, _ = np.linalg.qr(g((n, k)))
        a = (q1 * np.repeat(rng.random(max(1, k // 8) + 1) + 0.1, 8)[:k]) @ q2.conj().T
    # the sparsity a block of a composed tensor has: zero columns / rows (sectors one factor does not hold), and a product of
    # block-sparse factors (the R of its QR is a staircase with numerically zero rows in the middle)
    u = rng.random()
    if u < 0.2:
        a[:, rng.random(n) < rng.random()] = 0.0
    elif u < 0.3:
        a[rng.random(m) < rng.random()] = 0.0
    elif u < 0.4 and k >= 4:
        r = max(2, k // 2)
        b1, b2 = g((m, r)), g((r, n))
        b1[rng.random((m, 1)) < 0.5 * np.ones((1, r)) * (np.arange(r) % 2)] = 0.0
        b2[:, rng.random(n) < 0.4] = 0.0
        b2[np.arange(r) % 3 == 0, : n // 2] = 0.0
        a = b1 @ b2
    elif u < 0.48 and n >= 2:                         # exact copies of columns (dependence in the middle that is not a zero column)
        src = rng.integers(0, n, max(1, n // 5))
        dst = rng.integers(0, n, max(1, n // 5))
        a[:, dst] = a[:, src] * (rng.integers(1, 4, len(src)) if rng.random() < 0.5 else 1)
    elif u < 0.56:                                    # a scaled partial permutation (exactly orthogonal rows, ties, zero rows) + a few dense rows
        a = np.zeros((m, n), dtype=a.dtype)
        rows = rng.permutation(m)[:k]
        cols = rng.permutation(n)[:k]
        a[rows, cols] = rng.choice([1.0, 2.0, 0.5, 3.0], k) * (1 if rng.random() < 0.5 else g((k,)))
        a = a[:, :] * (rng.random(n) < 0.9)
        for r in rng.integers(0, m, int(rng.integers(0, 4))):
            a[r] = g((n,))
    return a * 10.0 ** int(rng.integers(-3, 4))


bad = 0
t0 = time.time()


def keep(tag, mats):
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    np.savez(f'gpurun_out/svd_fuzz_fail_seed{seed}_{tag}.npz', *mats)


for it in range(n_lists):
    cplx = it % 4 == 3
    mats = [rand_matrix(cplx) for _ in range(int(rng.integers(1, 9)))]
    try:
        res, info = bb.matrix_svd_batched([bb.as_block(a) for a in mats], return_info=True)
    except Exception as e:   # keep the list for a reproduction
        bad += 1
        keep(f'list{it}', mats)
        print(f'[fuzz] FAIL list {it} complex {cplx} shapes {[a.shape for a in mats]}: {e}', flush=True)
        continue
    for a, (u, s, vh), sw in zip(mats, res, info):
        u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
        k = min(a.shape)
        nrm = max(np.linalg.norm(a), 1e-300)
        e = [np.abs(s - np.linalg.svd(a, compute_uv=False)).max() / nrm, np.abs((u * s) @ vh - a).max() / nrm,
             np.abs(u.conj().T @ u - np.eye(k)).max(), np.abs(vh @ vh.conj().T - np.eye(k)).max()]
        if not (max(e) <= 1e-10 and np.all(s[:-1] >= s[1:] - 1e-10 * nrm) and sw <= 40):
            bad += 1
            keep(f'list{it}_{a.shape[0]}x{a.shape[1]}', [a])
            print(f'[fuzz] FAIL list {it} shape {a.shape} complex {cplx} sweeps {sw}: dS {e[0]:.1e} recon {e[1]:.1e} U {e[2]:.1e} V {e[3]:.1e}', flush=True)
    # QR of the same list (economic; every second list also mode='full') and the truncating caller's form of the SVD
    full = it % 2 == 1
    try:
        qrs = bb.matrix_qr_batched([bb.as_block(a) for a in mats], full)
    except Exception as e:   # (a verified fallback that gives up raises: loud, and counted)
        bad += 1
        keep(f'qr_list{it}', mats)
        print(f'[fuzz] FAIL qr list {it} complex {cplx} full {full} shapes {[a.shape for a in mats]}: {e}', flush=True)
        qrs = []
    for a, (q, r) in zip(mats, qrs):
        q, r = bb.to_numpy(q), bb.to_numpy(r)
        nrm = max(np.linalg.norm(a), 1e-300)
        e = [np.abs(q @ r - a).max() / nrm, np.abs(q.conj().T @ q - np.eye(q.shape[1])).max(), np.abs(np.tril(r, -1)).max() / nrm]
        if not max(e) <= 1e-10:
            bad += 1
            keep(f'qr_list{it}_{a.shape[0]}x{a.shape[1]}', [a])
            print(f'[fuzz] FAIL qr list {it} shape {a.shape} complex {cplx} full {full}: recon {e[0]:.1e} Q {e[1]:.1e} tril {e[2]:.1e}', flush=True)
    res2, ranks = bb.matrix_svd_batched([bb.as_block(a) for a in mats], null_vectors=False, return_rank=True)
    for a, (u, s, vh), rk in zip(mats, res2, ranks):
        u, s, vh = bb.to_numpy(u)[:, :rk], bb.to_numpy(s), bb.to_numpy(vh)[:rk]
        nrm = max(np.linalg.norm(a), 1e-300)
        tail = np.sqrt(np.sum(s[rk:] ** 2)) / nrm
        e = [np.abs(s - np.linalg.svd(a, compute_uv=False)).max() / nrm, np.abs((u * s[:rk]) @ vh - a).max() / nrm,
             np.abs(u.conj().T @ u - np.eye(rk)).max() if rk else 0.0, np.abs(vh @ vh.conj().T - np.eye(rk)).max() if rk else 0.0]
        if not (e[0] <= 1e-10 and e[1] <= 1e-10 + 10 * tail and max(e[2:]) <= 1e-10):
            bad += 1
            print(f'[fuzz] FAIL lazy svd list {it} shape {a.shape} complex {cplx} rank {rk}: dS {e[0]:.1e} recon {e[1]:.1e} (tail {tail:.1e}) U {e[2]:.1e} V {e[3]:.1e}', flush=True)
    if it % 2 == 1:   # a hermitian list: plain, Gram matrices of the blocks above (semi-definite, rank deficient, sparse), degenerate clusters,
        hs = []       # +- pairs, zero rows / columns
        for a in mats[:3]:
            n = int(rng.integers(2, 500))
            kind = int(rng.integers(0, 5))
            g = (lambda sh: rng.standard_normal(sh) + 1j * rng.standard_normal(sh)) if cplx else rng.standard_normal
            if kind == 0:
                z = g((n, n))
                h = z + z.conj().T
            elif kind == 1:
                b = a if max(a.shape) <= 600 else a[:600, :600]
                h = b @ b.conj().T if rng.random() < 0.5 else b.conj().T @ b
            elif kind == 2:
                q, _ = np.linalg.qr(g((n, n)))
                h = (q * np.repeat(rng.standard_normal(n // 6 + 1), 6)[:n]) @ q.conj().T
            elif kind == 3:
                q, _ = np.linalg.qr(g((n, n)))
                w = rng.random(n // 2 + 1)
                h = (q * np.concatenate([w, -w])[:n]) @ q.conj().T
            else:
                z = g((n, n))
                h = z + z.conj().T
                dead = rng.random(n) < 0.3
                h[dead] = 0.0
                h[:, dead] = 0.0
            hs.append((h + h.conj().T) / 2)
        for h, (w, v) in zip(hs, bb.eigh_batched([bb.as_block(h) for h in hs])):
            w, v = bb.to_numpy(w), bb.to_numpy(v)
            nrm = max(np.abs(h).max() * h.shape[0], 1e-300)
            e = [np.abs(w - np.linalg.eigvalsh(h)).max() / nrm, np.abs(h @ v - v * w).max() / nrm, np.abs(v.conj().T @ v - np.eye(h.shape[0])).max()]
            if not max(e) <= 1e-10:
                bad += 1
                keep(f'eigh_list{it}_{h.shape[0]}', [h])
                print(f'[fuzz] FAIL eigh list {it} n {h.shape[0]} complex {cplx}: dw {e[0]:.1e} resid {e[1]:.1e} V {e[2]:.1e}', flush=True)
    if it % 10 == 9:
        print(f'[fuzz] {it + 1} lists, {bad} failures, {time.time() - t0:.0f} s', flush=True)
print(f'[fuzz] done: {n_lists} lists, seed {seed}: {bad} failures')
sys.exit(1 if bad else 0)
