"""CPU restatement of the reference's abelian host loops around the per-block numpy calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain numpy + Python loops, ONE BLOCK AT A TIME
like the reference -- deliberately not sharing code with ``cyten_amd.abelian`` (which builds
grouped launches).  Tensors are plain data: an object with attributes
``moduli, legs[k].sectors/.mults/.sign, block_inds, blocks`` (e.g. ``cyten_amd.workloads.TensorSpec``).

Leg convention (same as the workload generators): every leg carries a sign (+1 incoming,
-1 outgoing) and a block is allowed iff sum_k sign_k q_k = 0 (mod the moduli) -- the flat form of
cyten's "fuse(codomain) == fuse(domain)" rule with legs = codomain + reversed(domain).
"""
import numpy as np

from . import block_ops as ops


def _reduce(q, moduli):
    q = np.array(q, dtype=np.int64, copy=True)
    for k, m in enumerate(moduli):
        if m:
            q[..., k] %= m
    return q


def leg_slices(leg):
    return np.concatenate([[0], np.cumsum(leg.mults)]).astype(int)


def to_dense(t):
    """Dense array of a block-sparse tensor (what ``Tensor.to_numpy()`` gives the reference tests)."""
    sl = [leg_slices(l) for l in t.legs]
    dtype = np.result_type(np.float64, *[np.asarray(b).dtype for b in t.blocks])   # complex blocks give a complex array
    out = np.zeros([int(s[-1]) for s in sl], dtype=dtype)
    for row, blk in zip(t.block_inds, t.blocks):
        out[tuple(slice(sl[k][i], sl[k][i + 1]) for k, i in enumerate(row))] = blk
    return out


def compose(a, b, num_contr):
    """``abelian_compose_worker`` (/root/reference/src/backends/abelian.cpp:1239-1469).

    Contract a.legs[-1-i] with b.legs[i], i < num_contr.  Returns (res_blocks, res_block_inds,
    n_matrix_dot): blocks in the order the reference creates them (col_b outer, row_a inner),
    then lexsorted.  Steps follow the reference line by line:
      key packing :1265-1283, lexsort of a :1286-1303, grouping :1305-1335, 2-D reshapes
      :1349-1382, coupled charges :1384-1418, charge lookup :1420, hot loop :1424-1460."""
    moduli = a.moduli
    na_keep = len(a.legs) - num_contr
    nb_keep = len(b.legs) - num_contr
    if len(a.blocks) == 0 or len(b.blocks) == 0:
        return [], np.zeros((0, na_keep + nb_keep), np.int64), 0
    a_bi = np.asarray(a.block_inds, dtype=np.int64)
    b_bi = np.asarray(b.block_inds, dtype=np.int64)
    a_keep, a_contr = a_bi[:, :na_keep], a_bi[:, na_keep:]
    b_contr, b_keep = b_bi[:, :num_contr], b_bi[:, num_contr:]
    nsecs = [len(b.legs[i].mults) for i in range(num_contr)]
    strides = [1]
    for i in range(1, num_contr):
        strides.append(strides[-1] * nsecs[i - 1])  # make_stride(cstyle=False)
    strides = np.array(strides[:num_contr], dtype=np.int64)
    a_keys = a_contr @ strides[::-1] if num_contr else np.zeros(len(a_bi), np.int64)
    b_keys = b_contr @ strides if num_contr else np.zeros(len(b_bi), np.int64)
    a_sort = np.lexsort(np.hstack([a_keys[:, None], a_keep]).T)
    a_keep, a_keys = a_keep[a_sort], a_keys[a_sort]
    a_blocks = [a.blocks[i] for i in a_sort]
    b_sort = np.lexsort(np.hstack([b_keys[:, None], b_keep]).T)  # b is lexsorted in the reference already
    b_keep, b_keys = b_keep[b_sort], b_keys[b_sort]
    b_blocks = [b.blocks[i] for i in b_sort]

    def row_diffs(keep):
        if keep.shape[1] == 0:
            return [0, keep.shape[0]]
        d = np.flatnonzero(np.any(keep[1:] != keep[:-1], axis=1)) + 1
        return [0] + d.tolist() + [keep.shape[0]]

    a_sl, b_sl = row_diffs(a_keep), row_diffs(b_keep)
    perm = list(range(num_contr - 1, -1, -1)) + list(range(num_contr, len(b.legs)))
    a_groups, b_groups, a_shape_keep, b_shape_keep = [], [], [], []
    for g in range(len(a_sl) - 1):
        blks = a_blocks[a_sl[g]:a_sl[g + 1]]
        shp = blks[0].shape[:na_keep]
        a_shape_keep.append(shp)
        a_groups.append([np.reshape(x, (int(np.prod(shp, dtype=np.int64)), -1)) for x in blks])
    for g in range(len(b_sl) - 1):
        blks = b_blocks[b_sl[g]:b_sl[g + 1]]
        shp = blks[0].shape[num_contr:]
        b_shape_keep.append(shp)
        b_groups.append([np.reshape(np.transpose(x, perm), (-1, int(np.prod(shp, dtype=np.int64)))) for x in blks])
    a_rows = a_keep[a_sl[:-1]]
    b_cols = b_keep[b_sl[:-1]]
    a_ch = np.zeros((len(a_rows), len(moduli)), np.int64)
    for k in range(na_keep):
        a_ch += a.legs[k].sign * a.legs[k].sectors[a_rows[:, k]]
    a_ch = _reduce(a_ch, moduli)
    b_ch = np.zeros((len(b_cols), len(moduli)), np.int64)
    for k in range(nb_keep):
        b_ch -= b.legs[num_contr + k].sign * b.legs[num_contr + k].sectors[b_cols[:, k]]
    b_ch = _reduce(b_ch, moduli)
    lookup = {}
    for r, ch in enumerate(map(tuple, a_ch)):
        lookup.setdefault(ch, []).append(r)

    res_blocks, res_rows, n_dot = [], [], 0
    for col_b in range(len(b_cols)):
        kb = b_keys[b_sl[col_b]:b_sl[col_b + 1]]
        for row_a in lookup.get(tuple(b_ch[col_b]), []):
            ka = a_keys[a_sl[row_a]:a_sl[row_a + 1]]
            # iter_common_sorted_1d: merge walk over two ascending key lists
            i = j = 0
            common = []
            while i < len(ka) and j < len(kb):
                if ka[i] < kb[j]:
                    i += 1
                elif ka[i] > kb[j]:
                    j += 1
                else:
                    common.append((i, j))
                    i += 1
                    j += 1
            if not common:
                continue
            k1, k2 = common[0]
            block = ops.matrix_dot(a_groups[row_a][k1], b_groups[col_b][k2])
            n_dot += 1
            for k1, k2 in common[1:]:
                block = block + ops.matrix_dot(a_groups[row_a][k1], b_groups[col_b][k2])
                n_dot += 1
            block = np.reshape(block, tuple(a_shape_keep[row_a]) + tuple(b_shape_keep[col_b]))
            res_blocks.append(block)
            res_rows.append(np.concatenate([a_rows[row_a], b_cols[col_b]]))
    if not res_blocks:
        return [], np.zeros((0, na_keep + nb_keep), np.int64), n_dot
    res_bi = np.array(res_rows, dtype=np.int64).reshape(len(res_rows), na_keep + nb_keep)
    order = np.lexsort(res_bi.T)
    return [res_blocks[i] for i in order], res_bi[order], n_dot


def fused_maps(moduli, legs, signs):
    """{coupled charge: [(sector-index tuple, offset, size)]} in C-style (lexicographic) order of
    the index tuples -- the sub-block layout of a combined leg (abelian.cpp:1022-1219)."""
    if not legs:
        return {tuple([0] * len(moduli)): [((), 0, 1)]}
    grids = np.indices([len(l.mults) for l in legs]).reshape(len(legs), -1).T
    q = np.zeros((grids.shape[0], len(moduli)), np.int64)
    sizes = np.ones(grids.shape[0], np.int64)
    for k, l in enumerate(legs):
        q += signs[k] * l.sectors[grids[:, k]]
        sizes *= l.mults[grids[:, k]]
    q = _reduce(q, moduli)
    out = {}
    for idx, ch, sz in zip(map(tuple, grids), map(tuple, q), sizes):
        lst = out.setdefault(ch, [])
        off = lst[-1][1] + lst[-1][2] if lst else 0
        lst.append((tuple(int(i) for i in idx), int(off), int(sz)))
    return out


def combine_to_matrix(t, num_codomain):
    """Fuse legs[:nc] -> rows, legs[nc:] -> columns: one zero-initialised 2-D block per coupled
    charge, every old block written into its sub-rectangle (abelian.cpp:1196-1217:
    ``bb.zeros`` + ``new_block[slices] = combined``).  Returns (charges, blocks, row_maps, col_maps)."""
    nc = num_codomain
    moduli = t.moduli
    rl, cl = t.legs[:nc], t.legs[nc:]
    rmap = fused_maps(moduli, rl, [l.sign for l in rl])
    cmap = fused_maps(moduli, cl, [-l.sign for l in cl])
    present = {}
    for bi, row in enumerate(np.asarray(t.block_inds)):
        q = np.zeros(len(moduli), np.int64)
        for k in range(nc):
            q += rl[k].sign * rl[k].sectors[row[k]]
        ch = tuple(int(x) for x in _reduce(q, moduli))
        present.setdefault(ch, []).append(bi)
    charges = sorted(present, key=lambda c: tuple(reversed(c)))
    blocks = []
    for ch in charges:
        rpos = {idx: (o, s) for idx, o, s in rmap[ch]}
        cpos = {idx: (o, s) for idx, o, s in cmap[ch]}
        big = np.zeros((sum(s for _, _, s in rmap[ch]), sum(s for _, _, s in cmap[ch])))
        for bi in present[ch]:
            row = t.block_inds[bi]
            ro, rs = rpos[tuple(int(i) for i in row[:nc])]
            co, cs = cpos[tuple(int(i) for i in row[nc:])]
            big[ro:ro + rs, co:co + cs] = np.reshape(t.blocks[bi], (rs, cs))
        blocks.append(big)
    return charges, blocks, [rmap[c] for c in charges], [cmap[c] for c in charges]


def svd_blocks(blocks, algorithm=None):
    """``AbelianBackend::svd`` loop (abelian.cpp:3499-3541): one ``matrix_svd`` per present block."""
    return [ops.matrix_svd(b, algorithm) for b in blocks]


def truncation_selection(S, qdims=None, chi_max=None, chi_min=1, degeneracy_tol=0.0, trunc_cut=0.0, svd_min=None,
                         minimize_error=True):
    """``TensorBackend::_truncate_singular_values_selection``
    (/root/reference/src/backends/tensor_backend.cpp:139-242), statement by statement."""
    S = np.asarray(S, dtype=float)
    marginal_errs = S ** 2 if qdims is None else qdims * S ** 2
    piv = np.argsort(marginal_errs, kind='stable')
    S = S[piv]
    marginal_errs = marginal_errs[piv]
    logS = np.log(np.choose(S <= 1.0e-100, [S, 1.0e-100 * np.ones(len(S))]))
    n = len(S)
    good = np.ones(n, dtype=np.bool_)

    def combine_constraints(good1, good2):
        res = np.logical_and(good1, good2)
        if np.any(res):
            return res
        return good1  # the reference warns and ignores the new constraint

    if chi_max is not None and chi_max < n:
        good2 = np.zeros(n, dtype=np.bool_)
        good2[-chi_max:] = True
        good = combine_constraints(good, good2)
    if chi_min > 1:
        good2 = np.ones(n, dtype=np.bool_)
        good2[-chi_min + 1:] = False
        good = combine_constraints(good, good2)
    if degeneracy_tol > 0:
        good2 = np.empty(n, np.bool_)
        good2[0] = True
        good2[1:] = np.greater_equal(logS[1:] - logS[:-1], degeneracy_tol)
        good = combine_constraints(good, good2)
    if svd_min is not None:
        good = combine_constraints(good, np.greater_equal(S, svd_min))
    good = combine_constraints(good, np.cumsum(marginal_errs) > trunc_cut * trunc_cut)
    nonzero = np.nonzero(good)[0]
    cut = int(nonzero[0]) if minimize_error else int(nonzero[-1])
    err = float(np.sum(marginal_errs[:cut]))
    new_norm = float(np.sum(marginal_errs[cut:]))
    mask = np.zeros(n, dtype=np.bool_)
    np.put(mask, piv[cut:], True)
    return mask, err, new_norm


def theta_tdot_svd(A, B, chi_max=None):
    """The BASELINE workload on the CPU, as the reference runs it: theta = compose(A, B, 1)
    (one np.dot per matched pair), combine to a matrix per coupled charge, one scipy SVD per
    block, host truncation.  Returns a dict with everything the parity tests compare."""
    blocks, bi, n_dot = compose(A, B, 1)

    class _T:
        pass

    th = _T()
    th.moduli = A.moduli
    th.legs = list(A.legs[:-1]) + list(B.legs[1:])
    th.block_inds, th.blocks = bi, blocks
    nc = len(A.legs) - 1
    charges, mats, rmaps, cmaps = combine_to_matrix(th, nc)
    usv = svd_blocks(mats)
    S_all = np.concatenate([s for _, s, _ in usv]) if usv else np.zeros(0)
    out = dict(theta_blocks=blocks, theta_block_inds=bi, n_matrix_dot=n_dot, charges=charges, matrices=mats,
               usv=usv, S_all=S_all, theta=th)
    if chi_max is not None:
        out['mask'], out['err'], out['new_norm'] = truncation_selection(S_all, chi_max=chi_max)
    return out
