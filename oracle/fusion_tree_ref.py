"""CPU restatement of the reference's FusionTreeBackend callers of the block backend, ONE numpy call per block-backend call
of the reference (TEST INFRASTRUCTURE, see oracle/__init__.py).  Spaces and data are plain objects with the attributes the
loops read: a space has ``sectors`` (sorted coupled sectors), ``multiplicities`` (block sizes), ``qdims``, ``tree_blocks[i]``
(list of objects with ``tree, start, stop, multiplicities``), ``num_legs`` and ``tree_block_slice(tree)``; data has
``block_inds`` (n, 2) and ``blocks`` -- e.g. cyten_amd.fusion_tree.TreeSpace / FusionTreeData with numpy blocks."""
import numpy as np

from . import block_ops as ops
from .abelian_ref import truncation_selection


def _common(a, b):
    where = {tuple(s): j for j, s in enumerate(np.asarray(b.sectors).tolist())}
    return [(i, where[tuple(s)]) for i, s in enumerate(np.asarray(a.sectors).tolist()) if tuple(s) in where]


def compose(a_inds, a_blocks, b_inds, b_blocks):
    """``FusionTreeBackend::compose`` (/root/reference/src/backends/fusion_tree_backend.cpp:669-698): ``iter_common_sorted_1d``
    over a's domain column and b's codomain column, one ``matrix_dot`` per hit"""
    blocks, rows = [], []
    a_inds, b_inds = np.asarray(a_inds).reshape(-1, 2), np.asarray(b_inds).reshape(-1, 2)
    i = j = 0
    while i < len(a_blocks) and j < len(b_blocks):
        if a_inds[i, 1] < b_inds[j, 0]:
            i += 1
        elif a_inds[i, 1] > b_inds[j, 0]:
            j += 1
        else:
            blocks.append(ops.matrix_dot(a_blocks[i], b_blocks[j]))
            rows.append((a_inds[i, 0], b_inds[j, 1]))
            i += 1
            j += 1
    return blocks, np.array(rows, dtype=np.int64).reshape(len(rows), 2)


def svd(a_inds, a_blocks, codomain, domain, new_mults=None, algorithm=None):
    """``::svd`` (:2184-2252): returns (u_blocks, u_inds), (s_blocks, s_inds), (vh_blocks, vh_inds)"""
    a_inds = np.asarray(a_inds).reshape(-1, 2)
    cm, dm = codomain.multiplicities, domain.multiplicities
    common = _common(codomain, domain)
    if new_mults is None:
        new_mults = [min(int(cm[i]), int(dm[j])) for i, j in common]
    u, ui, s, si, vh, vi = [], [], [], [], [], []
    n = 0
    for i_new, (i_cod, i_dom) in enumerate(common):
        ui.append((i_cod, i_new))
        vi.append((i_new, i_dom))
        if n < len(a_blocks) and a_inds[n, 0] == i_cod:
            uu, ss, vv = ops.matrix_svd(a_blocks[n], algorithm)
            u.append(uu), s.append(ss), vh.append(vv)
            si.append((i_new, i_new))
            n += 1
        else:
            u.append(np.eye(int(cm[i_cod]))[:, :int(new_mults[i_new])])
            vh.append(np.eye(int(dm[i_dom]))[:int(new_mults[i_new]), :])
    return (u, np.array(ui).reshape(-1, 2)), (s, np.array(si).reshape(-1, 2)), (vh, np.array(vi).reshape(-1, 2))


def qr(a_inds, a_blocks, codomain, domain, new_mults=None, lq=False):
    """``::qr`` (:2125-2180) / ``::lq`` (:2070-2123): ((iso blocks, inds), (triangular blocks, inds)) -- Q, R or (for lq) Q, L"""
    a_inds = np.asarray(a_inds).reshape(-1, 2)
    cm, dm = codomain.multiplicities, domain.multiplicities
    common = _common(codomain, domain)
    if new_mults is None:
        new_mults = [min(int(cm[i]), int(dm[j])) for i, j in common]
    q, qi, t, ti = [], [], [], []
    n = 0
    for i_new, (i_cod, i_dom) in enumerate(common):
        qi.append((i_new, i_dom) if lq else (i_cod, i_new))
        if n < len(a_blocks) and a_inds[n, 0] == i_cod:
            if lq:
                ll, qq = ops.matrix_lq(a_blocks[n], False)
                q.append(qq), t.append(ll)
                ti.append((i_cod, i_new))
            else:
                qq, rr = ops.matrix_qr(a_blocks[n], False)
                q.append(qq), t.append(rr)
                ti.append((i_new, i_dom))
            n += 1
        elif lq:
            q.append(np.eye(int(dm[i_dom]))[:int(new_mults[i_new]), :])
        else:
            q.append(np.eye(int(cm[i_cod]))[:, :int(new_mults[i_new])])
    return (q, np.array(qi).reshape(-1, 2)), (t, np.array(ti).reshape(-1, 2))


def eigh(a_inds, a_blocks, codomain, sort=None):
    """``::eigh`` (:2033-2067)"""
    a_inds = np.asarray(a_inds).reshape(-1, 2)
    w, v = [], []
    n = 0
    for i in range(len(codomain.multiplicities)):
        if n < len(a_blocks) and a_inds[n, 0] == i:
            ww, vv = ops.eigh(a_blocks[n], sort)
            w.append(ww), v.append(vv)
            n += 1
        else:
            v.append(np.eye(int(codomain.multiplicities[i])))
    return w, v


def truncate_singular_values(s_inds, s_blocks, domain, **options):
    """``::truncate_singular_values`` (:2254-2340): the dense S array over ALL sectors of the leg (zeros where S has no block),
    qdims repeated per sector, the selection of tensor_backend.cpp:139-242, then the Mask blocks of the sectors that keep
    something"""
    s_inds = np.asarray(s_inds).reshape(-1, 2)
    mults = [int(m) for m in domain.multiplicities]
    S_np = np.zeros(sum(mults))
    qd = np.empty(sum(mults))
    slices, stop, i = [], 0, 0
    for j, m in enumerate(mults):
        start, stop = stop, stop + m
        slices.append(slice(start, stop))
        if i < len(s_blocks) and s_inds[i, 0] == j:
            S_np[start:stop] = s_blocks[i]
            i += 1
        qd[start:stop] = domain.qdims[j]
    keep, err, new_norm = truncation_selection(S_np, qdims=qd, **options)
    blocks, rows = [], []
    for j, slc in enumerate(slices):
        blk = keep[slc]
        if not np.any(blk):
            continue
        rows.append((len(rows), j))
        blocks.append(blk)
    return blocks, np.array(rows, dtype=np.int64).reshape(len(rows), 2), err, new_norm


def transform_tensor(data_inds, data_blocks, codomain, domain, new_codomain, new_domain, codomain_idcs, domain_idcs, mapping):
    """``TreePairMapping::transform_tensor`` (/root/reference/src/backends/fusion_tree_mapping.cpp:391-513), statement by
    statement: axes :415-431, dtype :433-436, coupled sectors :441-450, tree-block pairs :453-454, terms :456-478
    (``get_item``, ``mul``, ``operator+``), old multiplicities :484-499, ``permute_combined_matrix`` + ``set_item`` :501-506."""
    data_inds = np.asarray(data_inds).reshape(-1, 2)
    J, K = codomain.num_legs, domain.num_legs
    N = J + K
    tree_block_axes_1 = [i if i < J else (N - 1) + (J - i) for i in codomain_idcs]
    tree_block_axes_2 = [i if i < J else (N - 1) + (J - i) for i in domain_idcs]
    leg_perm = list(codomain_idcs) + list(domain_idcs)[::-1]
    inv_leg_perm = list(np.argsort(leg_perm))
    cplx = any(np.iscomplexobj(b) for b in data_blocks) or any(
        isinstance(c, complex) and c.imag != 0.0 for tg in mapping.values() for c in tg.values())
    rows, blocks = [], []
    for i, j in _common(new_codomain, new_domain):
        block = np.zeros((int(new_codomain.multiplicities[i]), int(new_domain.multiplicities[j])), dtype=complex if cplx else float)
        is_zero_block = True
        for xb in new_codomain.tree_blocks[i]:
            for yb in new_domain.tree_blocks[j]:
                tree_block = None
                for (t1, t2), self_I in mapping.items():
                    coeff = self_I.get((xb.tree, yb.tree))
                    if coeff is None:
                        continue
                    _, b1 = codomain.tree_block_slice(t1)
                    jd, b2 = domain.tree_block_slice(t2)
                    which = np.flatnonzero(data_inds[:, 1] == jd)
                    if len(which) == 0:
                        continue
                    sub = data_blocks[int(which[0])][b1.start:b1.stop, b2.start:b2.stop]
                    add_block = coeff * sub
                    tree_block = add_block if tree_block is None else tree_block + add_block
                if tree_block is None:
                    continue
                is_zero_block = False
                leg_mults = list(xb.multiplicities) + list(yb.multiplicities)[::-1]
                old_mults = [leg_mults[k] for k in inv_leg_perm]
                old_mults_cod, old_mults_dom = old_mults[:J], old_mults[J:][::-1]
                block[xb.start:xb.stop, yb.start:yb.stop] = ops.permute_combined_matrix(
                    tree_block, old_mults_cod, tree_block_axes_1, old_mults_dom, tree_block_axes_2)
        if is_zero_block:
            continue
        rows.append((i, j))
        blocks.append(block)
    return blocks, np.array(rows, dtype=np.int64).reshape(len(rows), 2)
