"""Host side of the charge-block-sparse path: abelian sector bookkeeping around the grouped kernels.

cyten itself does not travel to the GPU box, so this module is the build's own counterpart of the
reference's ``AbelianBackend`` callers of the block backend:

* :func:`compose`  <- ``abelian_compose_worker`` (/root/reference/src/backends/abelian.cpp:1239-1469)
  The int64 sector matching (key packing, lexsort, grouping by kept legs, charge lookup, merge walk
  over contracted keys) is host work exactly as in the reference; the *hot loop* (:1424-1460) that
  issues one ``matrix_dot`` (+ ``operator+``) per matched pair is replaced by ONE grouped launch
  (`HipBlockBackend.make_gemm_plan`): each result block is one GEMM problem whose K-split pairs are
  accumulated inside the kernel.
* :func:`combine_legs_to_matrix` <- ``AbelianBackend::combine_legs`` (abelian.cpp:1022-1219):
  zero-fill + one batched strided scatter instead of ``zeros`` + ``set_item`` per block.
* :func:`svd` <- ``AbelianBackend::svd`` (abelian.cpp:3461-3568): one batched SVD over all sectors.
* :func:`truncate_singular_values` <- ``tensor_backend.cpp:139-242`` (host numpy, unchanged logic)
  + ``abelian.cpp:3623-3638``; the S blocks live in one device pool so the forced device->host
  transfer is a single copy.
* :func:`qr`, :func:`eigh` <- ``AbelianBackend::qr`` (:3084-3151), ``::eigh`` (:1759-1788).

The functions only need the block-backend *interface* (`matrix_dot_grouped`, `matrix_svd_batched`,
...), not a particular implementation.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Sequence

import numpy as np

__all__ = ['Symmetry', 'Leg', 'AbelianTensor', 'compose', 'compose_plan', 'compose_plan_py', 'combine_legs_to_matrix', 'svd',
           'truncate_singular_values', 'truncated_svd', 'qr', 'eigh', 'norm', 'inner', 'split_matrix_legs']


class Symmetry:
    """Product of U(1) (modulus 0) and Z_N (modulus N) factors; sectors are int vectors."""

    def __init__(self, moduli: Sequence[int]):
        self.moduli = tuple(int(m) for m in moduli)
        self.n = len(self.moduli)

    def reduce(self, q: np.ndarray) -> np.ndarray:
        q = np.array(q, dtype=np.int64, copy=True)
        for k, m in enumerate(self.moduli):
            if m:
                q[..., k] %= m
        return q

    def fuse(self, sector_lists, signs) -> np.ndarray:
        """Row-wise sum_k signs[k]*sector_lists[k] reduced by the moduli
        (``multiple_fusion_broadcast`` of abelian.cpp:1384-1418 for abelian groups)."""
        tot = np.zeros_like(np.asarray(sector_lists[0], dtype=np.int64))
        for s, sg in zip(sector_lists, signs):
            tot = tot + sg * np.asarray(s, dtype=np.int64)
        return self.reduce(tot)

    def __eq__(self, other):
        return isinstance(other, Symmetry) and other.moduli == self.moduli

    def __repr__(self):
        return 'Symmetry(' + ' x '.join('U(1)' if m == 0 else f'Z{m}' for m in self.moduli) + ')'


def _lexsort_rows(a: np.ndarray) -> np.ndarray:
    """``np.lexsort(a.T)``: last column is the primary key (BlockInds::lexsort_indices)."""
    if a.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    return np.lexsort(a.T)


class Leg:
    """ElementarySpace mirror: sorted sectors with multiplicities, and an orientation sign
    (+1: codomain-like / incoming charge, -1: domain-like / outgoing)."""

    def __init__(self, symmetry: Symmetry, sectors, mults, sign: int = +1):
        sectors = symmetry.reduce(np.asarray(sectors, dtype=np.int64).reshape(len(mults), symmetry.n))
        mults = np.asarray(mults, dtype=np.int64)
        order = _lexsort_rows(sectors)
        self.symmetry = symmetry
        self.sectors = sectors[order]
        self.mults = mults[order]
        self.sign = int(sign)
        self.slices = np.concatenate([[0], np.cumsum(self.mults)])

    @property
    def nsec(self):
        return len(self.mults)

    @property
    def dim(self):
        return int(self.mults.sum())

    def dual(self) -> 'Leg':
        return Leg(self.symmetry, self.sectors, self.mults, -self.sign)

    def can_contract_with(self, other: 'Leg') -> bool:
        return (self.sign == -other.sign and np.array_equal(self.sectors, other.sectors)
                and np.array_equal(self.mults, other.mults))

    def __repr__(self):
        return f'Leg(nsec={self.nsec}, dim={self.dim}, sign={self.sign:+d})'


@dataclass
class AbelianTensor:
    """AbelianBackendData mirror (/root/reference/include/cyten/backends/abelian.h:52-83): a list
    of dense blocks plus the int64 table ``block_inds`` (one row per block, one column per leg,
    entries = sector index on that leg), lexsorted.  Charge rule: sum_k sign_k * q_k = 0."""
    symmetry: Symmetry
    legs: list
    blocks: list
    block_inds: np.ndarray
    num_codomain: int = 0
    labels: list = field(default_factory=list)

    def __post_init__(self):
        self.block_inds = np.asarray(self.block_inds, dtype=np.int64).reshape(len(self.blocks), len(self.legs))

    @property
    def nlegs(self):
        return len(self.legs)

    def sorted(self) -> 'AbelianTensor':
        order = _lexsort_rows(self.block_inds)
        return AbelianTensor(self.symmetry, self.legs, [self.blocks[i] for i in order], self.block_inds[order],
                             self.num_codomain, self.labels)

    def block_shape(self, row) -> tuple:
        return tuple(int(l.mults[i]) for l, i in zip(self.legs, row))

    def check_charges(self):
        for row in self.block_inds:
            q = self.symmetry.fuse([l.sectors[i] for l, i in zip(self.legs, row)], [l.sign for l in self.legs])
            if np.any(q != 0):
                raise ValueError(f'block {row} violates the charge rule')

    @staticmethod
    def allowed_block_inds(symmetry, legs) -> np.ndarray:
        """All sector-index combinations with total charge 0 (lexsorted)."""
        grids = np.indices([l.nsec for l in legs]).reshape(len(legs), -1).T
        if grids.shape[0] == 0:
            return grids.astype(np.int64)
        q = symmetry.fuse([l.sectors[grids[:, k]] for k, l in enumerate(legs)], [l.sign for l in legs])
        ok = np.all(q == 0, axis=1)
        inds = grids[ok].astype(np.int64)
        return inds[_lexsort_rows(inds)]

    @classmethod
    def from_numpy_blocks(cls, bb, symmetry, legs, np_blocks, block_inds, num_codomain=0):
        return cls(symmetry, list(legs), [bb.as_block(b) for b in np_blocks], block_inds, num_codomain).sorted()

    @classmethod
    def from_spec(cls, bb, spec):
        """Upload a plain-data tensor (``cyten_amd.workloads.TensorSpec``: moduli, legs with
        sectors/mults/sign, block_inds, numpy blocks) to the device."""
        sym = Symmetry(spec.moduli)
        legs = [Leg(sym, l.sectors, l.mults, l.sign) for l in spec.legs]
        return cls.from_numpy_blocks(bb, sym, legs, spec.blocks, spec.block_inds, spec.num_codomain)

    def to_numpy_blocks(self, bb):
        return [bb.to_numpy(b) for b in self.blocks]

    def to_dense(self, bb) -> np.ndarray:
        """Dense array (test helper; the reference tests compare against ``.to_numpy()``)."""
        cplx = any(getattr(b, 'dtype', np.dtype('float64')).kind == 'c' for b in self.blocks)
        out = np.zeros([l.dim for l in self.legs], dtype=np.complex128 if cplx else np.float64)
        for blk, row in zip(self.blocks, self.block_inds):
            sl = tuple(slice(int(l.slices[i]), int(l.slices[i + 1])) for l, i in zip(self.legs, row))
            out[sl] = bb.to_numpy(blk)
        return out


# ---------------------------------------------------------------------------------------------
# compose / tdot
# ---------------------------------------------------------------------------------------------

@dataclass
class ComposePlan:
    """Result of the host-side sector matching: which (a-block, b-block) pairs feed which result
    block.  ``pairs[g]`` lists (index into a.blocks, index into b.blocks) for result block g."""
    res_block_inds: np.ndarray
    res_shapes: list
    pairs: list
    legs: list
    flops: float = 0.0


_native = None  # (lib, check) once libcyten_amd has been loaded; False if it cannot be


def _native_planner():
    global _native
    if _native is None:
        try:
            from . import _lib
            _native = (_lib.load(), _lib)
        except (ImportError, OSError):
            _native = False
    return _native


def _leg_descs(legs):
    """ctypes view of a leg list for the C++ planner (arrays kept alive by the returned tuple)."""
    _, L = _native
    arr = (L.LegDesc * max(len(legs), 1))()
    keep = []
    for i, lg in enumerate(legs):
        sec = np.ascontiguousarray(lg.sectors, dtype=np.int64)
        mul = np.ascontiguousarray(lg.mults, dtype=np.int64)
        keep += [sec, mul]
        arr[i].n_sectors, arr[i].sectors, arr[i].mults, arr[i].sign = lg.nsec, sec.ctypes.data, mul.ctypes.data, lg.sign
    return arr, keep


def compose_plan(a: AbelianTensor, b: AbelianTensor, num_contr: int) -> ComposePlan:
    """Sector matching of ``abelian_compose_worker`` (abelian.cpp:1265-1460): the C++ planner of
    ``csrc/abelian_plan.hip`` (``cyb_compose_plan_create``) when the library is built, else -- and as the specification
    the tests compare it with -- :func:`compose_plan_py`."""
    nat = _native_planner()
    if not nat or a.symmetry != b.symmetry:
        return compose_plan_py(a, b, num_contr)
    import ctypes as C
    lib, L = nat
    na_keep, nb_keep = a.nlegs - num_contr, b.nlegs - num_contr
    if na_keep < 0 or nb_keep < 0:
        return compose_plan_py(a, b, num_contr)
    res_legs = list(a.legs[:na_keep]) + list(b.legs[num_contr:])
    la, keep_a = _leg_descs(a.legs)
    lb, keep_b = _leg_descs(b.legs)
    abi = np.ascontiguousarray(a.block_inds, dtype=np.int64)
    bbi = np.ascontiguousarray(b.block_inds, dtype=np.int64)
    mod = np.array(a.symmetry.moduli, dtype=np.int64)
    handle = C.c_void_p()
    L.check(lib.cyb_compose_plan_create(mod.ctypes.data, a.symmetry.n, la, a.nlegs, abi.ctypes.data, len(a.blocks), lb, b.nlegs,
                                        bbi.ctypes.data, len(b.blocks), num_contr, C.byref(handle)))
    try:
        n_res, n_pairs, n_cols = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(lib.cyb_compose_plan_sizes(handle, C.byref(n_res), C.byref(n_pairs), C.byref(n_cols)))
        nr, npair, nc = n_res.value, n_pairs.value, n_cols.value
        res_bi = np.zeros((nr, nc), dtype=np.int64)
        shapes = np.zeros((nr, nc), dtype=np.int64)
        goff = np.zeros(nr + 1, dtype=np.int64)
        pa, pb = np.zeros(max(npair, 1), dtype=np.int64), np.zeros(max(npair, 1), dtype=np.int64)
        flops = C.c_double()
        L.check(lib.cyb_compose_plan_get(handle, res_bi.ctypes.data, shapes.ctypes.data, goff.ctypes.data, pa.ctypes.data,
                                         pb.ctypes.data, C.byref(flops)))
    finally:
        lib.cyb_compose_plan_destroy(handle)
    pal, pbl, gl = pa.tolist(), pb.tolist(), goff.tolist()
    pairs = [list(zip(pal[gl[g]:gl[g + 1]], pbl[gl[g]:gl[g + 1]])) for g in range(nr)]
    return ComposePlan(res_bi, [tuple(r) for r in shapes.tolist()], pairs, res_legs, flops.value)


def compose_plan_py(a: AbelianTensor, b: AbelianTensor, num_contr: int) -> ComposePlan:
    """Sector matching of ``abelian_compose_worker`` (abelian.cpp:1265-1460), int64 host work in numpy.

    Contracts the last `num_contr` legs of `a` with the first `num_contr` legs of `b`; as in the
    reference's leg layout (legs = codomain + reversed domain) a's contracted legs appear in
    REVERSED order relative to b's: ``a.legs[-1-i]`` pairs with ``b.legs[i]``."""
    na_keep = a.nlegs - num_contr
    for i in range(num_contr):
        if not a.legs[a.nlegs - 1 - i].can_contract_with(b.legs[i]):
            raise ValueError(f'legs a[{a.nlegs - 1 - i}] and b[{i}] are not contractible')
    res_legs = list(a.legs[:na_keep]) + list(b.legs[num_contr:])
    nb_keep = b.nlegs - num_contr
    empty = ComposePlan(np.zeros((0, na_keep + nb_keep), np.int64), [], [], res_legs)
    if len(a.blocks) == 0 or len(b.blocks) == 0:
        return empty
    a_keep, a_contr = a.block_inds[:, :na_keep], a.block_inds[:, na_keep:]
    b_contr, b_keep = b.block_inds[:, :num_contr], b.block_inds[:, num_contr:]
    # pack the contracted columns into one key, F-style strides over b's leg order (:1265-1283)
    nsecs = [b.legs[i].nsec for i in range(num_contr)]
    strides = np.ones(num_contr, dtype=np.int64)
    for i in range(1, num_contr):
        strides[i] = strides[i - 1] * nsecs[i - 1]
    a_keys = a_contr @ strides[::-1] if num_contr else np.zeros(len(a.blocks), np.int64)
    b_keys = b_contr @ strides if num_contr else np.zeros(len(b.blocks), np.int64)
    # sort a by (keep columns, contracted key): np.lexsort(hstack([key, keep]).T)  (:1286-1303)
    a_sort = _lexsort_rows(np.hstack([a_keys[:, None], a_keep]))
    a_keep, a_keys = a_keep[a_sort], a_keys[a_sort]
    # b is lexsorted already (last column primary) => grouped by its keep columns with ascending keys
    b_sort = _lexsort_rows(np.hstack([b_keys[:, None], b_keep]))
    b_keep, b_keys = b_keep[b_sort], b_keys[b_sort]

    def row_groups(keep):
        if keep.shape[1] == 0:
            return np.array([0, keep.shape[0]])
        diff = np.any(keep[1:] != keep[:-1], axis=1)
        return np.concatenate([[0], np.flatnonzero(diff) + 1, [keep.shape[0]]])

    a_sl, b_sl = row_groups(a_keep), row_groups(b_keep)
    a_rows, b_cols = a_keep[a_sl[:-1]], b_keep[b_sl[:-1]]
    # coupled charge of the kept legs of every row of a / column of b (:1384-1418)
    sym = a.symmetry
    if na_keep:
        a_ch = sym.fuse([a.legs[k].sectors[a_rows[:, k]] for k in range(na_keep)], [a.legs[k].sign for k in range(na_keep)])
    else:
        a_ch = np.zeros((len(a_rows), sym.n), np.int64)
    if nb_keep:
        b_ch = sym.fuse([b.legs[num_contr + k].sectors[b_cols[:, k]] for k in range(nb_keep)],
                        [-b.legs[num_contr + k].sign for k in range(nb_keep)])
    else:
        b_ch = np.zeros((len(b_cols), sym.n), np.int64)
    lookup: dict = {}
    for r, ch in enumerate(map(tuple, a_ch)):  # cyten.tools.misc.list_to_dict_list (:1420)
        lookup.setdefault(ch, []).append(r)

    res_rows, res_shapes, pairs = [], [], []
    flops = 0.0
    for cb in range(len(b_cols)):
        kb = b_keys[b_sl[cb]:b_sl[cb + 1]]
        for ra in lookup.get(tuple(b_ch[cb]), []):
            ka = a_keys[a_sl[ra]:a_sl[ra + 1]]
            common, ia, ib = np.intersect1d(ka, kb, assume_unique=True, return_indices=True)  # merge walk (:1430)
            if len(common) == 0:
                continue
            grp = [(int(a_sort[a_sl[ra] + i]), int(b_sort[b_sl[cb] + j])) for i, j in zip(ia, ib)]
            row = np.concatenate([a_rows[ra], b_cols[cb]])
            shp = tuple(int(res_legs[k].mults[row[k]]) for k in range(len(row)))
            res_rows.append(row)
            res_shapes.append(shp)
            pairs.append(grp)
            M = math.prod(map(int, shp[:na_keep]))
            N = math.prod(map(int, shp[na_keep:]))
            for ai, _ in grp:
                K = math.prod(map(int, a.block_shape(a.block_inds[ai])[na_keep:]))
                flops += 2.0 * M * N * K
    if not res_rows:
        return empty
    res_bi = np.array(res_rows, dtype=np.int64).reshape(len(res_rows), na_keep + nb_keep)
    order = _lexsort_rows(res_bi)
    return ComposePlan(res_bi[order], [res_shapes[i] for i in order], [pairs[i] for i in order], res_legs, flops)


def _compose_operands(bb, a, b, num_contr, plan):
    """2-D operand views for every block that takes part (reshape :1349-1382).  b-blocks need
    their contracted axes reversed; when that is not a stride-mergeable view all such blocks are
    made contiguous in ONE batched copy."""
    na_keep = a.nlegs - num_contr
    used_a = sorted({i for g in plan.pairs for i, _ in g})
    used_b = sorted({j for g in plan.pairs for _, j in g})
    a2 = {}
    a_src = bb.contiguous_many([a.blocks[i] for i in used_a])
    for i, blk in zip(used_a, a_src):
        rows = math.prod(map(int, blk.shape[:na_keep]))
        a2[i] = bb.reshape(blk, (rows, -1))
    perm = list(range(num_contr - 1, -1, -1)) + list(range(num_contr, b.nlegs))
    b_perm = [bb.permute_axes(b.blocks[j], perm) for j in used_b]
    b_perm = bb.contiguous_many(b_perm)  # no-op (no launch) when nothing was permuted
    b2 = {}
    for j, blk in zip(used_b, b_perm):
        cols = math.prod(map(int, blk.shape[num_contr:]))
        b2[j] = bb.reshape(blk, (-1, cols))
    return a2, b2


def make_compose_gemm(bb, a: AbelianTensor, b: AbelianTensor, num_contr: int, plan: ComposePlan | None = None):
    """Build the device launch plan of one contraction: returns (ComposePlan, GemmPlan)."""
    if plan is None:
        plan = compose_plan(a, b, num_contr)
    a2, b2 = _compose_operands(bb, a, b, num_contr, plan)
    groups = [[(a2[i], b2[j]) for i, j in g] for g in plan.pairs]
    return plan, (bb.make_gemm_plan(groups) if groups else None)


def compose(bb, a: AbelianTensor, b: AbelianTensor, num_contr: int) -> AbelianTensor:
    """Contract the last `num_contr` legs of a with the first `num_contr` legs of b."""
    plan = compose_plan(a, b, num_contr)
    na_keep = a.nlegs - num_contr
    if not plan.pairs:
        return AbelianTensor(a.symmetry, plan.legs, [], plan.res_block_inds, na_keep)
    a2, b2 = _compose_operands(bb, a, b, num_contr, plan)
    outs = bb.matrix_dot_grouped([[(a2[i], b2[j]) for i, j in g] for g in plan.pairs])
    blocks = [bb.reshape(o, shp) for o, shp in zip(outs, plan.res_shapes)]
    return AbelianTensor(a.symmetry, plan.legs, blocks, plan.res_block_inds, na_keep)


# ---------------------------------------------------------------------------------------------
# combine legs -> matrix, decompositions
# ---------------------------------------------------------------------------------------------

@dataclass
class MatrixView:
    """A tensor with its first `num_codomain` legs fused into a row leg and the rest into a column
    leg: one 2-D block per coupled charge (the form ``AbelianBackend::svd/qr/eigh`` work on)."""
    symmetry: Symmetry
    charges: np.ndarray      # (n_sectors, n_sym) coupled charge of each block
    blocks: list             # 2-D blocks (rows, cols)
    row_maps: list           # per sector: list of (leg-index tuple over the row legs, row slice start, size)
    col_maps: list
    row_legs: list
    col_legs: list


# Fusion maps and placement tables depend on the legs (and the block table) only, and the same combinations come back bond after
# bond, sweep after sweep: they are cached by CONTENT (the reference keeps the same tables inside its LegPipe objects,
# abelian.cpp:1022-1219, which live as long as the legs do).  Small bounded dictionaries, oldest entries dropped first.
_FUSE_CACHE: dict = {}
_PLACE_CACHE: dict = {}
_CACHE_MAX = 256


def _cache_put(cache, key, value):
    if len(cache) >= _CACHE_MAX:
        cache.pop(next(iter(cache)))
    cache[key] = value
    return value


def _legs_key(symmetry, legs, signs):
    return (symmetry.moduli,) + tuple((l.sectors.tobytes(), l.mults.tobytes(), int(sg)) for l, sg in zip(legs, signs))


def _fused_sector_maps(symmetry, legs, signs_override=None):
    """All sector-index combinations of `legs`, grouped by coupled charge:
    {charge tuple: [(index tuple, offset, size), ...]} in lexsorted (C-style) order
    (LegPipe fusion of abelian.cpp:1022-1219).  Cached; callers must not modify the result."""
    if not legs:
        return {tuple([0] * symmetry.n): [((), 0, 1)]}
    key = _legs_key(symmetry, legs, [l.sign for l in legs] if signs_override is None else signs_override)
    hit = _FUSE_CACHE.get(key)
    if hit is not None:
        return hit
    return _cache_put(_FUSE_CACHE, key, _fused_sector_maps_build(symmetry, legs, signs_override))


def _fused_sector_maps_build(symmetry, legs, signs_override=None):
    grids = np.indices([l.nsec for l in legs]).reshape(len(legs), -1).T
    signs = [l.sign for l in legs] if signs_override is None else signs_override
    q = symmetry.fuse([l.sectors[grids[:, k]] for k, l in enumerate(legs)], signs)
    sizes = np.prod([l.mults[grids[:, k]] for k, l in enumerate(legs)], axis=0)
    out: dict = {}
    for idx, ch, sz in zip(map(tuple, grids), map(tuple, q), sizes):
        lst = out.setdefault(ch, [])
        off = lst[-1][1] + lst[-1][2] if lst else 0
        lst.append((tuple(int(i) for i in idx), int(off), int(sz)))
    return out


def combine_legs_to_matrix(bb, t: AbelianTensor, num_codomain: int | None = None) -> MatrixView:
    """Fuse legs[:num_codomain] into rows and legs[num_codomain:] into columns.

    Reference: ``AbelianBackend::combine_legs`` allocates ``bb.zeros`` per result block and writes
    every old block with ``new_block[slices] = combined`` (abelian.cpp:1196-1217).  Here: ONE
    allocation + memset for the result block list and ONE batched strided scatter."""
    nc = t.num_codomain if num_codomain is None else num_codomain
    row_legs, col_legs = t.legs[:nc], t.legs[nc:]
    sym = t.symmetry
    binds = np.ascontiguousarray(t.block_inds, dtype=np.int64)
    # ---- placement table: which old block goes where in which coupled-charge matrix (legs and block table only: cached)
    key = (_legs_key(sym, row_legs, [l.sign for l in row_legs]), _legs_key(sym, col_legs, [-l.sign for l in col_legs]), nc,
           binds.shape, binds.tobytes())
    plan = _PLACE_CACHE.get(key)
    if plan is None:
        rmap = _fused_sector_maps(sym, row_legs)
        # column charge is defined so that row charge == column charge for an allowed block
        cmap = _fused_sector_maps(sym, col_legs, [-l.sign for l in col_legs])
        rpos = {ch: {idx: (off, sz) for idx, off, sz in lst} for ch, lst in rmap.items()}
        cpos = {ch: {idx: (off, sz) for idx, off, sz in lst} for ch, lst in cmap.items()}
        present: dict = {}
        if nc and len(binds):  # coupled charge of the row legs of every block at once
            ch_all = sym.fuse([l.sectors[binds[:, k]] for k, l in enumerate(row_legs)], [l.sign for l in row_legs]).tolist()
        else:
            ch_all = [[0] * sym.n] * len(binds)
        for bi, (row, ch) in enumerate(zip(binds.tolist(), ch_all)):
            present.setdefault(tuple(ch), []).append((bi, tuple(row[:nc]), tuple(row[nc:])))
        charges = sorted(present.keys(), key=lambda c: tuple(reversed(c)))
        n = len(binds)
        big_of, ro_a, co_a, rs_a, cs_a = (np.zeros(n, dtype=np.int64) for _ in range(5))
        for gi, ch in enumerate(charges):
            rp, cp = rpos[ch], cpos[ch]
            for bi, ridx, cidx in present[ch]:
                ro, rs = rp[ridx]
                co, cs = cp[cidx]
                big_of[bi], ro_a[bi], co_a[bi], rs_a[bi], cs_a[bi] = gi, ro, co, rs, cs
        shapes = [(sum(sz for _, _, sz in rmap[ch]), sum(sz for _, _, sz in cmap[ch])) for ch in charges]
        plan = _cache_put(_PLACE_CACHE, key, dict(
            charges=np.array(charges, dtype=np.int64).reshape(len(charges), sym.n), shapes=shapes, big_of=big_of, ro=ro_a, co=co_a,
            rs=rs_a, cs=cs_a, row_maps=[rmap[ch] for ch in charges], col_maps=[cmap[ch] for ch in charges]))
    shapes, big_of, ro_a, co_a, rs_a, cs_a = plan['shapes'], plan['big_of'], plan['ro'], plan['co'], plan['rs'], plan['cs']
    row_maps, col_maps = list(plan['row_maps']), list(plan['col_maps'])
    cplx = any(np.dtype(getattr(blk, 'dtype', np.float64)).kind == 'c' for blk in t.blocks)
    blocks = bb.zeros_many(shapes, dtype='complex128' if cplx else None)
    sub = getattr(bb, 'subblock', None)  # (a backend may offer the 2-D slice without the generality of get_item)
    fast = (hasattr(bb, 'copy_2d_many') and not cplx and len(binds) > 0
            and all(b.is_contiguous() and not b.is_bool for b in t.blocks))
    if fast:
        # placement as plain arrays (address, leading dimension, extents) per old block: one descriptor array filled by
        # numpy and one launch, no view objects per block (the 728-block U(1)xU(1) theta: 8 -> 2 ms of host time)
        base = np.array([b.ptr for b in blocks], dtype=np.int64)
        ld = np.array([sh[1] for sh in shapes], dtype=np.int64)
        dptr = base[big_of] + 8 * (ro_a * ld[big_of] + co_a)
        bb.copy_2d_many(dptr, ld[big_of], [b.ptr for b in t.blocks], cs_a, rs_a, cs_a)
    else:
        pairs = []
        for bi in range(len(binds)):
            big = blocks[int(big_of[bi])]
            ro, co, rs, cs = int(ro_a[bi]), int(co_a[bi]), int(rs_a[bi]), int(cs_a[bi])
            target = sub(big, ro, ro + rs, co, co + cs) if sub else bb.get_item(big, (slice(ro, ro + rs), slice(co, co + cs)))
            pairs.append((target, bb.reshape(t.blocks[bi], (rs, cs))))
        bb.copy_many(pairs)
    charges = plan['charges'].copy()
    return MatrixView(sym, charges, blocks, row_maps, col_maps,
                      list(row_legs), list(col_legs))


def svd(bb, mv: MatrixView, algorithm=None):
    """Thin SVD of every coupled-charge block in ONE batched call (abelian.cpp:3499-3541).
    Returns lists U, S, Vh (per sector)."""
    res = bb.matrix_svd_batched(mv.blocks, algorithm)
    return [r[0] for r in res], [r[1] for r in res], [r[2] for r in res]


def qr(bb, mv: MatrixView, full=False):
    res = bb.matrix_qr_batched(mv.blocks, full)
    return [r[0] for r in res], [r[1] for r in res]


def eigh(bb, mv: MatrixView, sort=None):
    res = bb.eigh_batched(mv.blocks, sort)
    return [r[0] for r in res], [r[1] for r in res]


def truncation_selection(S: np.ndarray, qdims=None, chi_max=None, chi_min=1, degeneracy_tol=0.0, trunc_cut=0.0,
                         svd_min=None, minimize_error=True):
    """Which singular values to keep: mirror of
    ``TensorBackend::_truncate_singular_values_selection`` (tensor_backend.cpp:139-242), pure host
    numpy like the reference.  Returns (mask, err, new_norm)."""
    S = np.asarray(S, dtype=np.float64)
    marginal = S ** 2 if qdims is None else np.asarray(qdims) * S ** 2
    piv = np.argsort(marginal, kind='stable')
    S_s, marg = S[piv], marginal[piv]
    logS = np.log(np.where(S_s <= 1e-100, 1e-100, S_s))
    n = len(S_s)
    good = np.ones(n, dtype=bool)

    def combine(good, good2):
        both = good & good2
        return both if both.any() else good  # keep the previous constraint set if incompatible

    if chi_max is not None and chi_max < n:
        g2 = np.zeros(n, dtype=bool)
        g2[-chi_max:] = True
        good = combine(good, g2)
    if chi_min > 1:
        g2 = np.ones(n, dtype=bool)
        g2[-chi_min + 1:] = False
        good = combine(good, g2)
    if degeneracy_tol > 0:
        g2 = np.empty(n, dtype=bool)
        g2[0] = True
        g2[1:] = (logS[1:] - logS[:-1]) >= degeneracy_tol
        good = combine(good, g2)
    if svd_min is not None:
        good = combine(good, S_s >= svd_min)
    good = combine(good, np.cumsum(marg) > trunc_cut * trunc_cut)
    nz = np.flatnonzero(good)
    cut = int(nz[0] if minimize_error else nz[-1])
    err = float(np.sum(marg[:cut]))
    new_norm = float(np.sum(marg[cut:]))
    mask = np.zeros(n, dtype=bool)
    mask[piv[cut:]] = True
    return mask, err, new_norm


def truncate_singular_values(bb, S_blocks, **options):
    """Pull all singular values to the host (the reference's forced sync point,
    abelian.cpp:3631), select, and return per-sector boolean masks + (err, new_norm)."""
    sizes = [s.size for s in S_blocks]
    if hasattr(bb, 'concatenate_to_numpy'):  # one gather launch + one download instead of one download per sector
        S_all = bb.concatenate_to_numpy(S_blocks)
    else:
        S_all = np.concatenate([bb.to_numpy(s) for s in S_blocks]) if S_blocks else np.zeros(0)
    mask, err, new_norm = truncation_selection(S_all, **options)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    return [mask[offs[i]:offs[i + 1]] for i in range(len(sizes))], err, new_norm


def truncated_svd(bb, theta: AbelianTensor, num_codomain=None, lazy_null=False, **options):
    """combine -> batched SVD -> truncation -> batched mask gather (decompositions.cpp:673-712).  With a backend that
    offers ``truncate_select`` the selection runs on the device and the gather reads the kept positions from there
    (the host sees the kept counts, err and new_norm only); otherwise -- and for lists beyond the device limit -- the
    reference's host selection on the downloaded singular values.

    ``lazy_null=True``: the singular vectors of numerically zero singular values are only computed if the truncation
    keeps them.  Every block of a two-site theta = A.B is rank-deficient, and ``svd_apply_mask`` (decompositions.cpp:620-631)
    throws those vectors away; the first SVD call skips their orthonormal completion (``CYB_SVD_SKIP_NULL_VECTORS``) and
    reports the numerical ranks, and only a sector whose kept count exceeds its rank -- ``chi_max`` beyond the number of
    non-zero singular values -- is decomposed again in full.  The returned factors are those of the eager path."""
    mv = combine_legs_to_matrix(bb, theta, num_codomain)
    ranks = None
    if lazy_null and hasattr(bb, 'lib'):
        res, ranks = bb.matrix_svd_batched(mv.blocks, null_vectors=False, return_rank=True)
        U, S, Vh = [r[0] for r in res], [r[1] for r in res], [r[2] for r in res]
    else:
        U, S, Vh = svd(bb, mv)
    masks = None
    if hasattr(bb, 'truncate_select') and options.get('qdims') is None and 0 < sum(s.size for s in S) <= bb.TRUNCATE_MAX:
        masks, _, err, new_norm = bb.truncate_select(S, **options)
    if masks is None:
        masks, err, new_norm = truncate_singular_values(bb, S, **options)
    if ranks is not None:   # sectors that keep a deflated singular value need their null vectors after all
        kept = [int(m.n if hasattr(m, 'n') else np.count_nonzero(m)) for m in masks]
        redo = [i for i, (k, r) in enumerate(zip(kept, ranks)) if k > r]
        if redo:
            full = bb.matrix_svd_batched([mv.blocks[i] for i in redo])
            for i, (u, s, vh) in zip(redo, full):
                U[i], S[i], Vh[i] = u, s, vh
    gathered = bb.mask_gather_many([(u, m, 1) for u, m in zip(U, masks)] + [(s, m, 0) for s, m in zip(S, masks)]
                                   + [(v, m, 0) for v, m in zip(Vh, masks)])
    n = len(U)
    return mv, gathered[:n], gathered[n:2 * n], gathered[2 * n:], err, new_norm


def split_matrix_legs(bb, mv: MatrixView, blocks, side: str):
    """Split the fused row ('rows': U-like blocks (rows, k)) or column ('cols': Vh-like (k, cols))
    leg back into the original legs (abelian.cpp:3414-3434 does get_item+reshape per block).
    Row slices are views; the column slices of all sectors are made contiguous by ONE batched gather."""
    if side not in ('rows', 'cols'):
        raise ValueError(f"side must be 'rows' or 'cols', got {side!r}")
    out, subs = [], []
    for sec, blk in enumerate(blocks):
        maps = mv.row_maps[sec] if side == 'rows' else mv.col_maps[sec]
        legs = mv.row_legs if side == 'rows' else mv.col_legs
        for idx, off, sz in maps:
            dims = [int(l.mults[i]) for l, i in zip(legs, idx)]
            if side == 'rows':
                sub = bb.get_item(blk, (slice(off, off + sz), slice(None)))
                out.append((sec, idx, bb.reshape(sub, dims + [blk.shape[1]])))
            else:
                subs.append(bb.get_item(blk, (slice(None), slice(off, off + sz))))
                out.append((sec, idx, [blk.shape[0]] + dims))
    if side == 'cols':
        dense = bb.contiguous_many(subs)
        out = [(sec, idx, bb.reshape(d, shape)) for (sec, idx, shape), d in zip(out, dense)]
    return out


def norm(bb, t: AbelianTensor) -> float:
    """abelian.cpp:2781-2792: one reduction over the whole block list."""
    return bb.norm_many(t.blocks)


def inner(bb, a: AbelianTensor, b: AbelianTensor) -> float:
    """abelian.cpp:2159-2211: <a|b> over the blocks present in both (same legs)."""
    ia = {tuple(r): i for i, r in enumerate(a.block_inds)}
    xs, ys = [], []
    for j, r in enumerate(b.block_inds):
        i = ia.get(tuple(r))
        if i is not None:
            xs.append(a.blocks[i])
            ys.append(b.blocks[j])
    return bb.inner_many(xs, ys)


# ---------------------------------------------------------------------------------------------
# leg permutation and vector-space operations (what the Krylov solvers need between composes)
# ---------------------------------------------------------------------------------------------

def permute_legs(bb, t: AbelianTensor, perm: Sequence[int], num_codomain: int | None = None) -> AbelianTensor:
    """Reorder the legs (AbelianBackend::permute_legs, abelian.cpp:2860-2905, without leg bending:
    signs stay with their legs).  Blocks become strided views (no data movement here: the next
    compose makes the operands it needs contiguous in one batched copy), the block table is
    re-sorted."""
    perm = [int(p) for p in perm]
    if sorted(perm) != list(range(t.nlegs)):
        raise ValueError(f'permute_legs: {perm} is not a permutation of {t.nlegs} legs')
    legs = [t.legs[p] for p in perm]
    blocks = [bb.permute_axes(b, perm) for b in t.blocks]
    out = AbelianTensor(t.symmetry, legs, blocks, t.block_inds[:, perm] if len(blocks) else t.block_inds.reshape(0, t.nlegs),
                        t.num_codomain if num_codomain is None else num_codomain)
    return out.sorted()


def _align(a: AbelianTensor, b: AbelianTensor):
    if a.nlegs != b.nlegs or any(x.nsec != y.nsec or x.sign != y.sign for x, y in zip(a.legs, b.legs)):
        raise ValueError('tensors live on different legs')
    ia = {tuple(r): i for i, r in enumerate(a.block_inds)}
    ib = {tuple(r): j for j, r in enumerate(b.block_inds)}
    both = [(ia[k], ib[k]) for k in ia if k in ib]
    only_a = [ia[k] for k in ia if k not in ib]
    only_b = [ib[k] for k in ib if k not in ia]
    return both, only_a, only_b


def linear_combination(bb, alpha: float, a: AbelianTensor, beta: float, b: AbelianTensor) -> AbelianTensor:
    """alpha*a + beta*b (abelian.cpp:2254-2302): blocks present in both go through ONE axpby launch,
    blocks present in only one of them are scaled copies (one more launch per side, if any)."""
    both, only_a, only_b = _align(a, b)
    rows, blocks = [], []
    if both:
        outs = bb.linear_combination_many(alpha, [a.blocks[i] for i, _ in both], beta, [b.blocks[j] for _, j in both])
        rows += [a.block_inds[i] for i, _ in both]
        blocks += outs
    if only_a:
        rows += [a.block_inds[i] for i in only_a]
        blocks += bb.mul_many(alpha, [a.blocks[i] for i in only_a])
    if only_b:
        rows += [b.block_inds[j] for j in only_b]
        blocks += bb.mul_many(beta, [b.blocks[j] for j in only_b])
    bi = np.array(rows, dtype=np.int64).reshape(len(rows), a.nlegs)
    return AbelianTensor(a.symmetry, a.legs, blocks, bi, a.num_codomain).sorted()


def scale(bb, alpha: float, a: AbelianTensor) -> AbelianTensor:
    """alpha * a, one launch over the block list (abelian.cpp:2230-2252)."""
    return AbelianTensor(a.symmetry, a.legs, bb.mul_many(alpha, a.blocks), a.block_inds, a.num_codomain)


def tdot(bb, a: AbelianTensor, b: AbelianTensor, legs_a: Sequence[int], legs_b: Sequence[int]) -> AbelianTensor:
    """``cyten.tdot(a, b, legs_a, legs_b)`` (tensors.py / abelian.cpp:1239-1469 behind it): contract leg
    ``legs_a[i]`` of a with leg ``legs_b[i]`` of b; the result carries a's remaining legs followed by b's
    remaining legs, each in their original order.  = two leg permutations (views) + one ``compose``."""
    legs_a = [int(i) % a.nlegs for i in legs_a]
    legs_b = [int(i) % b.nlegs for i in legs_b]
    if len(legs_a) != len(legs_b) or len(set(legs_a)) != len(legs_a) or len(set(legs_b)) != len(legs_b):
        raise ValueError('tdot: legs_a and legs_b must list the same number of distinct legs')
    keep_a = [i for i in range(a.nlegs) if i not in legs_a]
    keep_b = [i for i in range(b.nlegs) if i not in legs_b]
    # compose pairs a.legs[-1 - i] with b.legs[i]: a's contracted legs go last in REVERSED order
    a_p = permute_legs(bb, a, keep_a + legs_a[::-1])
    b_p = permute_legs(bb, b, legs_b + keep_b)
    return compose(bb, a_p, b_p, len(legs_a))
