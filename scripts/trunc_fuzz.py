"""Randomised soak of the device truncation (`truncate_select`, csrc/truncation.hip) against the host mirror of
``_truncate_singular_values_selection`` (abelian.truncation_selection, tensor_backend.cpp:139-242): random sector lists of singular
values -- graded, ties inside and across sectors, exact zeros, one-value sectors -- and random option sets (chi_max, chi_min,
degeneracy_tol, trunc_cut, svd_min, minimize_error; also incompatible ones, which the reference resolves by keeping the earlier
constraint).  Compared: the boolean mask, the kept positions per sector, err and new_norm.
`python scripts/trunc_fuzz.py [n_rounds=2000] [seed=0]`"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from cyten_amd import abelian as ab
from cyten_amd.block_backend import HipBlockBackend

n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(seed)
bad = 0
t0 = time.time()


def rand_values(n):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        s = rng.random(n)
    elif kind == 1:
        s = np.logspace(0, -float(rng.integers(2, 17)), n) * (1 + 0.01 * rng.random(n))
    elif kind == 2:                                   # ties: a few distinct values, repeated
        s = rng.choice(rng.random(max(1, n // 4)) + 0.01, n)
    elif kind == 3:                                   # exact zeros at the end
        s = rng.random(n)
        s[rng.random(n) < 0.4] = 0.0
    else:                                             # nearly degenerate multiplets (degeneracy_tol matters)
        s = np.repeat(rng.random(n // 3 + 1) + 0.05, 3)[:n] * (1 + 1e-9 * rng.standard_normal(n))
    return np.sort(np.abs(s))[::-1].copy()


for it in range(n_rounds):
    n_sec = int(rng.integers(1, 12))
    sizes = [int(rng.integers(1, 2 if rng.random() < 0.2 else 200)) for _ in range(n_sec)]
    S = [rand_values(n) * 10.0 ** int(rng.integers(-2, 3)) for n in sizes]
    if rng.random() < 0.3:                            # the same values in several sectors: ties across sectors
        for k in range(1, n_sec):
            if rng.random() < 0.5:
                m = min(sizes[k], sizes[0])
                S[k][:m] = S[0][:m]
                S[k] = np.sort(S[k])[::-1].copy()
    n = sum(sizes)
    o = {}
    if rng.random() < 0.8:
        o['chi_max'] = int(rng.integers(1, n + 3))
    if rng.random() < 0.3:
        o['chi_min'] = int(rng.integers(1, n + 2))
    if rng.random() < 0.3:
        o['degeneracy_tol'] = float(10.0 ** rng.integers(-10, -1))
    if rng.random() < 0.4:
        o['trunc_cut'] = float(10.0 ** rng.integers(-12, 1))
    if rng.random() < 0.3:
        o['svd_min'] = float(10.0 ** rng.integers(-12, 1))
    if rng.random() < 0.2:
        o['minimize_error'] = False
    S_all = np.concatenate(S)
    mask_h, err_h, nn_h = ab.truncation_selection(S_all, **o)
    tables, mask_d, err_d, nn_d = bb.truncate_select([bb.as_block(s) for s in S], **o)
    mask_d = bb.to_numpy(mask_d).astype(bool)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    ok = True
    if not np.array_equal(mask_h, mask_d):
        # ties at the cut: which member of a group of EQUAL values is kept is a matter of the (stable) sort order; the count and
        # the kept values must agree
        ok = mask_h.sum() == mask_d.sum() and np.array_equal(np.sort(S_all[mask_h]), np.sort(S_all[mask_d]))
    sc = max(np.sum(S_all ** 2), 1e-300)
    if not (ok and abs(err_h - err_d) <= 1e-12 * sc and abs(nn_h - nn_d) <= 1e-12 * sc):
        bad += 1
        print(f'[trunc-fuzz] FAIL round {it} sizes {sizes} opts {o}: kept {mask_h.sum()} vs {mask_d.sum()}, err {err_h} vs {err_d}, '
              f'new_norm {nn_h} vs {nn_d}', flush=True)
        continue
    for k, t in enumerate(tables):                    # the per-sector position tables are the device mask, ascending
        want = np.flatnonzero(mask_d[offs[k]:offs[k + 1]])
        got = bb.ctx.d2h_ptr(t.ptr, t.n, np.int64) if hasattr(bb.ctx, 'd2h_ptr') else None
        if t.n != len(want) or (got is not None and not np.array_equal(got, want)):
            bad += 1
            print(f'[trunc-fuzz] FAIL round {it} sector {k}: table {t.n} entries vs {len(want)}', flush=True)
            break
    if it % 500 == 499:
        print(f'[trunc-fuzz] {it + 1} rounds, {bad} failures, {time.time() - t0:.0f} s', flush=True)
print(f'[trunc-fuzz] done: {n_rounds} rounds, seed {seed}: {bad} failures')
sys.exit(1 if bad else 0)
