"""Host side of the non-abelian path: the block-backend callers of the reference's ``FusionTreeBackend`` on its own
data structure.

The symmetry layer -- fusion trees, F / R / B symbols, `TreePairMapping::from_instructions` -- is integer and symbol algebra
that never touches block data and stays what it is in the reference ("host-side fusion-tree bookkeeping stays").  What this
module mirrors is everything BETWEEN that layer and the block backend:

* :class:`FusionTreeData`  <- ``FusionTreeData`` (include/cyten/backends/fusion_tree_backend.h): one 2-D block per coupled
  sector, ``block_inds[n] = (i, j)``: index of the coupled sector in the codomain's / domain's ``sector_decomposition``.
* :class:`TreeSpace`  <- the part of ``TensorProduct`` the callers read: ``sector_decomposition``, ``block_size(i)``
  (``tp_mults``), ``sector_qdims``, ``iter_tree_blocks`` / ``tree_block_slice`` (which rows of a coupled block belong to which
  fusion tree, and the multiplicities of its uncoupled sectors).
* :func:`compose`  <- ``FusionTreeBackend::compose`` (src/backends/fusion_tree_backend.cpp:669-698): one ``matrix_dot`` per
  common coupled sector, no accumulation -- here ONE grouped launch.
* :func:`svd` / :func:`qr` / :func:`lq` / :func:`eigh`  <- ``::svd`` (:2184-2252), ``::qr`` (:2125-2180), ``::lq`` (:2070-2123),
  ``::eigh`` (:2033-2067): one batched decomposition over the present blocks, slices of the identity for the sectors
  without a block.
* :func:`truncate_singular_values`  <- ``::truncate_singular_values`` (:2254-2340): the selection of
  tensor_backend.cpp:139-242 with the marginal errors weighted by the sector's quantum dimension -- on the device
  (``cyb_truncate_select_weighted_f64``), the host sees counts, err and new_norm.
* :func:`transform_tensor`  <- ``TreePairMapping::transform_tensor`` (src/backends/fusion_tree_mapping.cpp:391-513), the body
  of ``FusionTreeBackend::apply_instructions`` (:593-631): the mapping (coefficients between tree pairs) comes from the
  symmetry layer as data; the loops over coupled sectors and tree pairs are the reference's, but instead of ``zeros`` +
  (``get_item``, ``mul``, ``operator+``) per term + ``permute_combined_matrix`` + ``set_item`` per tree pair they fill ONE
  descriptor list for ``HipBlockBackend.transform_blocks`` (one zero fill + one launch per tensor, complex coefficients).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

__all__ = ['TreeBlock', 'TreeSpace', 'FusionTreeData', 'compose', 'svd', 'qr', 'lq', 'eigh', 'truncate_singular_values',
           'transform_tensor', 'discard_zero_blocks', 'norm']


@dataclass(frozen=True)
class TreeBlock:
    """one fusion tree of a coupled sector: its rows inside the coupled block and the multiplicities of its uncoupled
    sectors (``TreeBlockInfo`` of ``TensorProduct::iter_tree_blocks``: tree, slice, multiplicities)"""
    tree: object           # hashable identifier of the fusion tree (the symmetry layer's FusionTree, or any key)
    start: int
    stop: int
    multiplicities: tuple


@dataclass
class TreeSpace:
    """What the block path reads of a ``TensorProduct``: sorted coupled sectors, block size and quantum dimension of each,
    and the tree blocks inside each coupled sector (in ``iter_tree_blocks`` order)."""
    sectors: np.ndarray            # (n_coupled, n_sym) int64: sector_decomposition, lexsorted
    qdims: np.ndarray              # (n_coupled,) sector_qdims
    tree_blocks: list              # per coupled sector: [TreeBlock, ...] with contiguous ascending slices
    num_legs: int = 0              # num_flat_legs
    _where: dict = field(default_factory=dict, repr=False)

    def __post_init__(self):
        self.sectors = np.asarray(self.sectors, dtype=np.int64).reshape(len(self.tree_blocks), -1)
        self.qdims = np.asarray(self.qdims, dtype=np.float64)
        self._where = {}
        for i, tbs in enumerate(self.tree_blocks):
            for tb in tbs:
                self._where[tb.tree] = (i, tb)

    @property
    def num_sectors(self):
        return len(self.tree_blocks)

    @property
    def multiplicities(self) -> np.ndarray:
        """block_size(i) of every coupled sector (``tp_mults``)"""
        return np.array([tbs[-1].stop if tbs else 0 for tbs in self.tree_blocks], dtype=np.int64)

    def block_size(self, i: int) -> int:
        tbs = self.tree_blocks[i]
        return tbs[-1].stop if tbs else 0

    def tree_block_slice(self, tree):
        """(coupled sector index, TreeBlock) of a tree (``tree_block_slice`` + the tree's ``coupled``)"""
        return self._where[tree]

    @classmethod
    def from_multiplicities(cls, sectors, tree_mults, qdims=None, num_legs=0, names=None):
        """``tree_mults[i]``: list of multiplicity tuples of the trees of coupled sector i (block rows in that order)"""
        blocks = []
        for i, lst in enumerate(tree_mults):
            off, tbs = 0, []
            for t, m in enumerate(lst):
                m = tuple(int(x) for x in m)
                sz = int(np.prod(m)) if m else 1
                tbs.append(TreeBlock(names[i][t] if names is not None else (i, t), off, off + sz, m))
                off += sz
            blocks.append(tbs)
        q = np.ones(len(blocks)) if qdims is None else qdims
        return cls(sectors, q, blocks, num_legs)


def common_sectors(a: TreeSpace, b: TreeSpace):
    """(i, j) of the coupled sectors both spaces hold, ascending (``iter_common_sorted_arrays`` of the two sorted
    ``sector_decomposition``s)"""
    where = {tuple(s): j for j, s in enumerate(b.sectors.tolist())}
    return [(i, where[tuple(s)]) for i, s in enumerate(a.sectors.tolist()) if tuple(s) in where]


@dataclass
class FusionTreeData:
    block_inds: np.ndarray     # (n_blocks, 2), lexsorted, duplicate free
    blocks: list

    def __post_init__(self):
        self.block_inds = np.asarray(self.block_inds, dtype=np.int64).reshape(len(self.blocks), 2)

    def sorted(self) -> 'FusionTreeData':
        order = np.lexsort(self.block_inds.T) if len(self.blocks) else np.zeros(0, dtype=np.int64)
        return FusionTreeData(self.block_inds[order], [self.blocks[i] for i in order])

    def block_ind_from_domain_sector(self, j: int):
        """index of the block whose domain sector index is j (``block_ind_from_coupled``), or None"""
        hit = np.flatnonzero(self.block_inds[:, 1] == j)
        return int(hit[0]) if len(hit) else None


# ---------------------------------------------------------------------------------------------------------------------------

def compose(bb, a: FusionTreeData, b: FusionTreeData) -> FusionTreeData:
    """fusion_tree_backend.cpp:669-698: blocks of a and b that share the coupled sector (a's domain index == b's codomain
    index) are multiplied, nothing is accumulated.  One grouped launch for the tensor."""
    ja = {int(j): n for n, j in enumerate(a.block_inds[:, 1])}
    groups, rows = [], []
    for m, i in enumerate(b.block_inds[:, 0]):
        n = ja.get(int(i))
        if n is not None:
            groups.append([(a.blocks[n], b.blocks[m])])
            rows.append((int(a.block_inds[n, 0]), int(b.block_inds[m, 1])))
    if not groups:
        return FusionTreeData(np.zeros((0, 2), np.int64), [])
    return FusionTreeData(np.array(rows, dtype=np.int64), bb.matrix_dot_grouped(groups)).sorted()


def _eye_slice(bb, dim, rows, cols, like):
    eye = bb.eye_matrix(int(dim), dtype=getattr(like, 'dtype', None))
    return bb.get_item(eye, (slice(0, rows) if rows is not None else slice(None), slice(0, cols) if cols is not None else slice(None)))


def _decompose(bb, a: FusionTreeData, codomain: TreeSpace, domain: TreeSpace, new_mults, kind, arg=None):
    """shared loop of ::svd / ::qr / ::lq: every common coupled sector gets its isometries, the sectors without a block from
    the identity; ONE batched call for the present blocks"""
    common = common_sectors(codomain, domain)
    have = {int(i): n for n, i in enumerate(a.block_inds[:, 0])}
    cm, dm = codomain.multiplicities, domain.multiplicities
    if new_mults is None:
        new_mults = [min(int(cm[i]), int(dm[j])) for i, j in common]
    present = [(k, i, j) for k, (i, j) in enumerate(common) if i in have]
    srcs = [a.blocks[have[i]] for _, i, _ in present]
    like = srcs[0] if srcs else None
    if kind == 'svd':
        facs = bb.matrix_svd_batched(srcs, arg) if srcs else []
    elif kind == 'qr':
        facs = bb.matrix_qr_batched(srcs, False) if srcs else []
    else:
        facs = bb.matrix_lq_batched(srcs, False) if srcs else []
    it = iter(facs)
    done = {k for k, _, _ in present}
    return common, new_mults, done, it, like


def svd(bb, a: FusionTreeData, codomain: TreeSpace, domain: TreeSpace, new_mults=None, algorithm=None):
    """fusion_tree_backend.cpp:2184-2252.  Returns (U, S, Vh) as FusionTreeData with block_inds (i_cod, i_new), (i_new, i_new),
    (i_new, i_dom); S only for the sectors that have a block."""
    common, new_mults, done, it, like = _decompose(bb, a, codomain, domain, new_mults, 'svd', algorithm)
    cm, dm = codomain.multiplicities, domain.multiplicities
    ub, ur, sb, sr, vb, vr = [], [], [], [], [], []
    for k, (i, j) in enumerate(common):
        ur.append((i, k))
        vr.append((k, j))
        if k in done:
            u, s, vh = next(it)
            ub.append(u), sb.append(s), vb.append(vh)
            sr.append((k, k))
        else:
            ub.append(_eye_slice(bb, cm[i], None, int(new_mults[k]), like))
            vb.append(_eye_slice(bb, dm[j], int(new_mults[k]), None, like))
    return FusionTreeData(ur, ub), FusionTreeData(sr, sb), FusionTreeData(vr, vb)


def qr(bb, a: FusionTreeData, codomain: TreeSpace, domain: TreeSpace, new_mults=None):
    """fusion_tree_backend.cpp:2125-2180: (Q with rows (i_cod, i_new), R with rows (i_new, i_dom))"""
    common, new_mults, done, it, like = _decompose(bb, a, codomain, domain, new_mults, 'qr')
    cm = codomain.multiplicities
    qb, qrw, rb, rr = [], [], [], []
    for k, (i, j) in enumerate(common):
        qrw.append((i, k))
        if k in done:
            q, r = next(it)
            qb.append(q), rb.append(r)
            rr.append((k, j))
        else:
            qb.append(_eye_slice(bb, cm[i], None, int(new_mults[k]), like))
    return FusionTreeData(qrw, qb), FusionTreeData(rr, rb)


def lq(bb, a: FusionTreeData, codomain: TreeSpace, domain: TreeSpace, new_mults=None):
    """fusion_tree_backend.cpp:2070-2123: (L with rows (i_cod, i_new), Q with rows (i_new, i_dom))"""
    common, new_mults, done, it, like = _decompose(bb, a, codomain, domain, new_mults, 'lq')
    dm = domain.multiplicities
    lb, lr, qb, qrw = [], [], [], []
    for k, (i, j) in enumerate(common):
        qrw.append((k, j))
        if k in done:
            l, q = next(it)
            lb.append(l), qb.append(q)
            lr.append((i, k))
        else:
            qb.append(_eye_slice(bb, dm[j], int(new_mults[k]), None, like))
    return FusionTreeData(lr, lb), FusionTreeData(qrw, qb)


def eigh(bb, a: FusionTreeData, codomain: TreeSpace, sort=None):
    """fusion_tree_backend.cpp:2033-2067: (W, V); a sector without a block has eigenvalues 0 (no W block) and the standard
    basis as eigenvectors"""
    have = {int(i): n for n, i in enumerate(a.block_inds[:, 0])}
    srcs = [a.blocks[have[i]] for i in sorted(have)]
    res = iter(bb.eigh_batched(srcs, sort) if srcs else [])
    like = srcs[0] if srcs else None
    vb, wb = [], []
    for i in range(codomain.num_sectors):
        if i in have:
            w, v = next(res)
            wb.append(w), vb.append(v)
        else:
            vb.append(bb.eye_matrix(codomain.block_size(i), dtype=getattr(like, 'dtype', None)))
    return (FusionTreeData(a.block_inds[np.argsort(a.block_inds[:, 0], kind='stable')].copy(), wb),
            FusionTreeData([(i, i) for i in range(codomain.num_sectors)], vb))


def truncate_singular_values(bb, S: FusionTreeData, domain: TreeSpace, **options):
    """fusion_tree_backend.cpp:2254-2340: every sector of the new leg contributes ``multiplicities[j]`` values (zeros where S
    has no block) weighted by ``sector_qdims[j]``.  Returns (mask_blocks, mask_block_inds, err, new_norm): boolean host
    vectors of the sectors that keep at least one value, rows (small index, large index j), as the reference builds its
    Mask.  The selection runs on the device when the backend offers it (weights are one number per sector)."""
    mults = [int(m) for m in domain.multiplicities]
    have = {int(i): n for n, i in enumerate(S.block_inds[:, 0])}
    absent = [j for j in range(len(mults)) if j not in have and mults[j] > 0]
    zeros = iter(bb.zeros_many([(mults[j],) for j in absent])) if absent else iter(())
    blocks = []
    for j, m in enumerate(mults):
        if m == 0:
            continue
        blocks.append(S.blocks[have[j]] if j in have else next(zeros))
    sec = [j for j, m in enumerate(mults) if m > 0]
    q = np.array([domain.qdims[j] for j in sec], dtype=np.float64)
    if hasattr(bb, 'truncate_select') and 0 < sum(mults) <= bb.TRUNCATE_MAX:
        _, mask, err, new_norm = bb.truncate_select(blocks, qdims=q, **options)
        keep = bb.to_numpy(mask).astype(bool)
    else:
        from . import abelian as ab
        S_np = np.concatenate([bb.to_numpy(b) for b in blocks]) if blocks else np.zeros(0)
        keep, err, new_norm = ab.truncation_selection(S_np, qdims=np.repeat(q, [mults[j] for j in sec]), **options)
    out_b, out_r, off = [], [], 0
    for j in sec:
        blk = keep[off:off + mults[j]]
        off += mults[j]
        if blk.any():
            out_r.append((len(out_r), j))
            out_b.append(blk.copy())
    return out_b, np.array(out_r, dtype=np.int64).reshape(len(out_r), 2), float(err), float(new_norm)


def norm(bb, a: FusionTreeData, codomain: TreeSpace) -> float:
    """fusion_tree_backend.cpp:1283-1297: sqrt(sum_n qdim(coupled_n) |block_n|^2), one reduction per block list"""
    if not a.blocks:
        return 0.0
    q = codomain.qdims[a.block_inds[:, 0]]
    if np.all(q == q[0]):
        return float(np.sqrt(q[0]) * bb.norm_many(a.blocks))
    return float(np.sqrt(sum(float(qi) * bb.norm_many([b]) ** 2 for qi, b in zip(q, a.blocks))))


# ---------------------------------------------------------------------------------------------------------------------------

def transform_tensor(bb, data: FusionTreeData, codomain: TreeSpace, domain: TreeSpace, new_codomain: TreeSpace, new_domain: TreeSpace,
                     codomain_idcs, domain_idcs, mapping) -> FusionTreeData:
    """``TreePairMapping::transform_tensor`` (fusion_tree_mapping.cpp:391-513).  `mapping`: the data of the symmetry layer's
    TreePairMapping, ``{(old codomain tree, old domain tree): {(new codomain tree, new domain tree): coefficient}}``
    (``mapping.data[I][J]``, f(T)_{Jm} = sum_I mapping[I][J] T_{Im}, :401).  `codomain_idcs` / `domain_idcs`: which old legs
    (numbered codomain first, then the domain legs counted from the end, :415-431) make up the new codomain / domain.

    Loops as in the reference -- common coupled sectors of the new spaces (:441-450), new tree-block pairs (:453-454), the old
    tree pairs that feed them (:456-478), old multiplicities in the new axis order (:484-499) -- but what they emit is one
    update record per new tree pair; the arithmetic is ONE ``transform_blocks`` call.  The result is complex if the data or a
    coefficient is (:433-436)."""
    J, K = codomain.num_legs, domain.num_legs
    N = J + K
    axes1 = [i if i < J else (N - 1) + (J - i) for i in codomain_idcs]
    axes2 = [i if i < J else (N - 1) + (J - i) for i in domain_idcs]
    leg_perm = list(codomain_idcs) + list(domain_idcs)[::-1]
    inv_leg_perm = [0] * len(leg_perm)
    for pos, src in enumerate(leg_perm):
        inv_leg_perm[src] = pos
    by_new: dict = {}        # (new tree pair) -> [(old codomain tree, old domain tree, coefficient)]
    for (t1, t2), targets in mapping.items():
        for new_pair, coeff in targets.items():
            by_new.setdefault(new_pair, []).append((t1, t2, coeff))
    block_of_dom = {int(j): n for n, j in enumerate(data.block_inds[:, 1])}
    new_rows, shapes, updates = [], [], []
    for i, j in common_sectors(new_codomain, new_domain):
        ups = []
        for xb in new_codomain.tree_blocks[i]:
            for yb in new_domain.tree_blocks[j]:
                srcs = by_new.get((xb.tree, yb.tree))
                if not srcs:
                    continue
                terms = []
                for t1, t2, coeff in srcs:
                    _, b1 = codomain.tree_block_slice(t1)
                    jd, b2 = domain.tree_block_slice(t2)
                    n = block_of_dom.get(jd)
                    if n is None:          # the old tensor has no block in that coupled sector
                        continue
                    terms.append((coeff, n, (b1.start, b1.stop), (b2.start, b2.stop)))
                if not terms:
                    continue
                leg_mults = list(xb.multiplicities) + list(yb.multiplicities)[::-1]
                old_mults = [leg_mults[k] for k in inv_leg_perm]
                ups.append((len(shapes), (xb.start, xb.stop), (yb.start, yb.stop), old_mults[:J], axes1, old_mults[J:][::-1], axes2, terms))
        if not ups:
            continue                       # (is_zero_block, :509-511)
        updates += ups
        new_rows.append((i, j))
        shapes.append((new_codomain.block_size(i), new_domain.block_size(j)))
    if not shapes:
        return FusionTreeData(np.zeros((0, 2), np.int64), [])
    return FusionTreeData(np.array(new_rows, dtype=np.int64), bb.transform_blocks(data.blocks, shapes, updates))


def discard_zero_blocks(bb, data: FusionTreeData, eps: float) -> FusionTreeData:
    """``FusionTreeData::discard_zero_blocks`` after ``apply_instructions`` (fusion_tree_backend.cpp:629): blocks whose
    largest entry is at most `eps` are dropped -- ONE batched reduction decides for the whole tensor when the backend has one"""
    if not data.blocks:
        return data
    keep = [n for n, b in enumerate(data.blocks) if bb.max_abs(b) > eps]
    if len(keep) == len(data.blocks):
        return data
    return FusionTreeData(data.block_inds[keep], [data.blocks[n] for n in keep])
