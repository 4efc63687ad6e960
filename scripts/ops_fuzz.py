"""Randomised soak of the data-movement / BLAS-1 / GEMM side of the operator API against numpy: random shapes, permuted and
sliced views, real / complex / mixed operands.  `python scripts/ops_fuzz.py [n_rounds=200] [seed=0]`"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from cyten_amd.block_backend import HipBlockBackend

n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(seed)
bad = 0


def rnd(shape, cplx):
    a = rng.standard_normal(shape)
    return a + 1j * rng.standard_normal(shape) if cplx else a


def view_of(a_np):
    """a device view and the matching numpy view: the block itself, a permutation, a slice, or both"""
    blk, ref = bb.as_block(a_np), a_np
    if ref.ndim >= 2 and rng.random() < 0.5:
        perm = list(rng.permutation(ref.ndim))
        blk, ref = bb.permute_axes(blk, perm), ref.transpose(perm)
    if rng.random() < 0.4:
        key = tuple(slice(int(rng.integers(0, max(1, d // 3))), int(d - rng.integers(0, max(1, d // 3))), int(rng.integers(1, 3))) for d in ref.shape)
        blk, ref = bb.get_item(blk, key), ref[key]
    return blk, ref


def check(tag, got, want, tol=1e-12):
    global bad
    got = np.asarray(got)
    want = np.asarray(want)
    scale = max(1.0, float(np.abs(want).max()) if want.size else 1.0)
    if got.shape != want.shape or (want.size and not np.abs(got - want).max() <= tol * scale):
        bad += 1
        err = np.abs(got - want).max() if got.shape == want.shape and want.size else float('nan')
        print(f'[ops-fuzz] FAIL {tag}: shapes {got.shape} / {want.shape} err {err:.2e}', flush=True)


t0 = time.time()
for it in range(n_rounds):
    cplx = bool(rng.random() < 0.35)
    nd = int(rng.integers(1, 5))
    shape = tuple(int(x) for x in rng.integers(1, 14 if nd > 2 else 60, nd))
    a_np, b_np = rnd(shape, cplx), rnd(shape, bool(rng.random() < 0.35))
    a, ar = view_of(a_np)
    # copies of views, elementwise, BLAS-1
    check('contiguous', bb.to_numpy(bb.contiguous(a)), ar, 0.0)
    check('mul', bb.to_numpy(bb.mul(0.37, a)), 0.37 * ar)
    b = bb.as_block(np.ascontiguousarray(b_np.reshape(-1)[:ar.size].reshape(ar.shape)) if b_np.size >= ar.size else rnd(ar.shape, False))
    br = bb.to_numpy(b)
    check('linear_combination', bb.to_numpy(bb.linear_combination(1.5, a, -0.25, b)), 1.5 * ar - 0.25 * br)
    check('norm', bb.norm(a), np.linalg.norm(ar.ravel()), 1e-12)
    check('inner', bb.inner(a, b, True), np.vdot(ar.ravel(), br.ravel()), 1e-11)
    check('abs', bb.to_numpy(bb.abs(a)), np.abs(ar))
    check('conj', bb.to_numpy(bb.conj(a)), np.conj(ar), 0.0)
    if ar.ndim >= 2:
        ax = int(rng.integers(0, ar.ndim))
        f = rng.standard_normal(ar.shape[ax])
        check('scale_axis', bb.to_numpy(bb.scale_axis(a, bb.as_block(f), ax)), ar * f.reshape([-1 if k == ax else 1 for k in range(ar.ndim)]))
        mask = rng.random(ar.shape[ax]) < 0.6
        check('apply_mask', bb.to_numpy(bb.apply_mask(a, mask, ax)), np.compress(mask, ar, axis=ax), 0.0)
        check('sum', bb.to_numpy(bb.sum(a, ax)), ar.sum(axis=ax), 1e-11)
    # reshape of a contiguous copy, set_item
    c = bb.contiguous(a)
    if ar.size > 1:
        check('reshape', bb.to_numpy(bb.reshape(c, (ar.size,))), ar.reshape(-1), 0.0)
    tgt_np = rnd(ar.shape, cplx or np.iscomplexobj(ar))
    tgt = bb.as_block(tgt_np.copy())
    key = tuple(slice(int(rng.integers(0, max(1, d // 2))), d) for d in ar.shape)
    val_np = rnd(tgt_np[key].shape, np.iscomplexobj(tgt_np))
    bb.set_item(tgt, key, bb.as_block(val_np))
    tgt_np[key] = val_np
    check('set_item', bb.to_numpy(tgt), tgt_np, 0.0)
    # grouped GEMM with K-split groups and transposed operands; tdot over random axes
    groups, refs = [], []
    for _ in range(int(rng.integers(1, 6))):
        M, N = (int(x) for x in rng.integers(1, 200, 2))
        g, ref = [], 0
        gc = bool(rng.random() < 0.3)
        for _ in range(int(rng.integers(1, 4))):
            K = int(rng.integers(1, 150))
            x_np, y_np = rnd((M, K), gc), rnd((K, N), gc and rng.random() < 0.7)
            if rng.random() < 0.5:
                x = bb.permute_axes(bb.as_block(np.ascontiguousarray(x_np.T)), [1, 0])
            else:
                x = bb.as_block(x_np)
            if rng.random() < 0.5:
                y = bb.permute_axes(bb.as_block(np.ascontiguousarray(y_np.T)), [1, 0])
            else:
                y = bb.as_block(y_np)
            g.append((x, y))
            ref = ref + x_np @ y_np
        groups.append(g)
        refs.append(ref)
    for o, ref in zip(bb.matrix_dot_grouped(groups), refs):
        check('matrix_dot_grouped', bb.to_numpy(o), ref, 1e-11)
    if ar.ndim >= 2:
        k = int(rng.integers(1, ar.ndim + 1))
        axes_a = [int(x) for x in rng.permutation(ar.ndim)[:k]]
        other = rnd(tuple(ar.shape[i] for i in axes_a) + (int(rng.integers(1, 9)),), cplx)
        check('tdot', bb.to_numpy(bb.tdot(a, bb.as_block(other), axes_a, list(range(k)))), np.tensordot(ar, other, (axes_a, list(range(k)))), 1e-11)
    if it % 50 == 49:
        print(f'[ops-fuzz] {it + 1} rounds, {bad} failures, {time.time() - t0:.0f} s', flush=True)
print(f'[ops-fuzz] done: {n_rounds} rounds, seed {seed}: {bad} failures')
sys.exit(1 if bad else 0)
