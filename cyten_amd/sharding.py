"""Sector sharding across the GPUs of one node.

Units of work are independent per output block: every result block of a tdot is one GEMM problem,
every coupled-charge block is one SVD/QR/eigh.  The reference has no multi-device notion
(all blocks of a tensor live on one device, abelian.h:39-44); this is new design:

* partition: greedy longest-processing-time (LPT) bin packing of the units by algorithmic flops;
* layout: all results of a sharded op live in one flat *pool* laid out rank-major
  ``[rank 0 segment | rank 1 segment | ...]`` with equal (padded) segment length, every unit at a
  fixed offset inside its owner's segment.  Each rank's kernels write straight into its segment
  (no packing pass), and ONE ``all_gather_into_tensor`` (RCCL over xGMI on GPUs, gloo on CPU)
  leaves the complete block list addressable on every rank -- which the unchanged host code
  expects -- as views into the pool (no unpacking pass either).

The module only uses torch.distributed and integer arithmetic; the per-unit compute is supplied
by the caller.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def lpt_assign(costs, n_bins: int) -> np.ndarray:
    """Greedy LPT: units sorted by descending cost go to the currently lightest bin.
    Returns owner[u] in [0, n_bins). Deterministic (ties broken by unit index)."""
    costs = np.asarray(costs, dtype=np.float64)
    owner = np.zeros(len(costs), dtype=np.int64)
    load = np.zeros(n_bins, dtype=np.float64)
    for u in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = int(np.argmin(load))  # first lightest bin
        owner[u] = r
        load[r] += costs[u]
    return owner


@dataclass
class PoolLayout:
    """Where every unit's elements live in the rank-major pool."""
    owner: np.ndarray      # (n_units,) rank owning unit u
    offset: np.ndarray     # (n_units,) element offset of unit u in the WHOLE pool
    sizes: np.ndarray      # (n_units,) elements
    seg_len: int           # padded per-rank segment length (elements)
    world: int

    @property
    def total(self) -> int:
        return self.seg_len * self.world

    def local_units(self, rank: int):
        return [int(u) for u in np.flatnonzero(self.owner == rank)]

    def imbalance(self, costs) -> float:
        """max over ranks of the assigned cost / mean cost (1.0 = perfect)."""
        costs = np.asarray(costs, dtype=np.float64)
        load = np.array([costs[self.owner == r].sum() for r in range(self.world)])
        return float(load.max() / max(load.mean(), 1e-300))


def make_layout(sizes, costs, world: int, align: int = 32) -> PoolLayout:
    """LPT-assign units (by `costs`) and lay them out rank-major; unit offsets are multiples of
    `align` elements (256-byte alignment for fp64)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    if world == 1:  # nothing to balance: units in order, offsets by one cumulative sum
        padded = (sizes + align - 1) // align * align
        offset = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.int64) if len(sizes) else np.zeros(0, np.int64)
        return PoolLayout(np.zeros(len(sizes), dtype=np.int64), offset, sizes, int(max(int(padded.sum()), align)), 1)
    owner = lpt_assign(costs, world)
    local_off = np.zeros(len(sizes), dtype=np.int64)
    seg = np.zeros(world, dtype=np.int64)
    for u in range(len(sizes)):
        r = owner[u]
        local_off[u] = seg[r]
        seg[r] += (sizes[u] + align - 1) // align * align
    seg_len = int(max(int(seg.max()) if len(seg) else 0, align))
    offset = owner * seg_len + local_off
    return PoolLayout(owner, offset, sizes, seg_len, world)


def allgather_pool(pool, layout: PoolLayout, rank: int, group=None):
    """One collective: afterwards every rank holds every rank's segment.  `pool` is a flat torch
    tensor of ``layout.total`` elements whose ``rank`` segment was written by this rank."""
    import torch.distributed as dist
    if layout.world == 1 and not (dist.is_available() and dist.is_initialized()):
        return pool
    seg = pool[rank * layout.seg_len:(rank + 1) * layout.seg_len]
    if pool.is_cuda:
        # NCCL/RCCL in-place form: the input is this rank's chunk of the output
        dist.all_gather_into_tensor(pool, seg, group=group)
    else:
        dist.all_gather_into_tensor(pool, seg.clone(), group=group)
    return pool


def layout_for_owner(sizes, owner, world: int, align: int = 32) -> PoolLayout:
    """Rank-major pool layout for units whose owner is already decided (the factors of a sector live with the rank that
    decomposed it): unit offsets are multiples of `align` elements inside the owner's segment."""
    sizes = np.asarray(sizes, dtype=np.int64)
    owner = np.asarray(owner, dtype=np.int64)
    local_off = np.zeros(len(sizes), dtype=np.int64)
    seg = np.zeros(world, dtype=np.int64)
    for u in range(len(sizes)):
        r = owner[u]
        local_off[u] = seg[r]
        seg[r] += (sizes[u] + align - 1) // align * align
    seg_len = int(max(int(seg.max()) if len(seg) else 0, align))
    return PoolLayout(owner, owner * seg_len + local_off, sizes, seg_len, world)


@dataclass
class SectorPlan:
    """Sharding of one theta = tdot(A, B) -> truncated SVD step by COUPLED CHARGE (last bullet): the sector
    partition of theta equals that of the combined matrix, so a rank that owns a sector contracts exactly the theta blocks
    that land in it and theta never has to be gathered."""
    sector_of_block: np.ndarray   # (n_theta_blocks,) sector index of every result block of the contraction
    shapes: list                  # (rows, cols) of every sector's combined matrix
    costs: np.ndarray             # nominal SVD flops 4 m n^2 + 8 n^3 per sector (the shard weights)
    layout: PoolLayout            # sector -> rank (LPT by cost); sizes = min(rows, cols)
    s_layout: PoolLayout          # rank-major pool of the singular values (one all_gather makes S global)

    def blocks_of(self, sectors) -> np.ndarray:
        """theta blocks (indices into the contraction plan, ascending = block-table order) of the given sectors"""
        return np.flatnonzero(np.isin(self.sector_of_block, np.asarray(list(sectors), dtype=np.int64)))


def theta_sector_plan(plan, a, num_codomain: int, world: int) -> SectorPlan:
    """Group the result blocks of a contraction plan (``abelian.ComposePlan``) by the coupled charge of their first
    `num_codomain` legs -- the sectors ``combine_legs_to_matrix`` produces, in its order -- and LPT-assign the sectors to
    `world` ranks by nominal SVD flops.  Pure int64 host work."""
    from . import abelian as ab
    sym = a.symmetry
    legs = plan.legs
    nc = num_codomain
    binds = np.asarray(plan.res_block_inds, dtype=np.int64)
    row_legs, col_legs = legs[:nc], legs[nc:]
    ch = sym.fuse([l.sectors[binds[:, k]] for k, l in enumerate(row_legs)], [l.sign for l in row_legs])
    keys = [tuple(c) for c in ch.tolist()]
    charges = sorted(set(keys), key=lambda c: tuple(reversed(c)))
    index = {c: i for i, c in enumerate(charges)}
    sector_of_block = np.array([index[k] for k in keys], dtype=np.int64)
    rmap = ab._fused_sector_maps(sym, row_legs)
    cmap = ab._fused_sector_maps(sym, col_legs, [-l.sign for l in col_legs])
    shapes = [(sum(sz for _, _, sz in rmap[c]), sum(sz for _, _, sz in cmap[c])) for c in charges]
    costs = np.array([4.0 * max(s) * min(s) ** 2 + 8.0 * min(s) ** 3 for s in shapes])
    ks = np.array([min(s) for s in shapes], dtype=np.int64)
    layout = make_layout(ks, costs, world)
    s_layout = layout_for_owner(ks, layout.owner, world)
    return SectorPlan(sector_of_block, shapes, costs, layout, s_layout)


# ---------------------------------------------------------------------------------------------------------------------
# Splitting ONE block's Jacobi iteration over ranks ("unless the largest blocks are themselves split")
#
# The single-device engine (csrc/jacobi_engine.hip) already splits every block pair of a round over G workgroups BY
# COLUMNS: each part owns whole 64-column chunks of the pair's rows of W (and of J), computes a partial 32 x 32 Gram
# matrix, the parts exchange their partials, every part solves the same small eigenproblem and updates its own columns.
# The multi-device form is the same decomposition one level up: rank r owns a column slab of W and J of the WHOLE block
# for the whole iteration, so rows never move between ranks; per round one all_reduce sums the partial Gram matrices of
# all pairs of the round (pairs x 32 x 32 doubles: 196 KB for the 1442-row block of the chi=4096 list).  The functions
# below are the ownership / schedule logic and an executable specification of the algorithm on torch.distributed (numpy
# arithmetic; gloo on CPU in tests/test_distributed.py).  Whether it PAYS is a latency question, answered in DESIGN.md
# section 5: a round is ~40 us of which only ~19 us (the MFMA share) shrink with the number of ranks, and an all_reduce
# over xGMI costs about as much as it saves.
# ---------------------------------------------------------------------------------------------------------------------

JB = 16          # rows per block of the engine (csrc/jacobi_engine.h)
CHUNK = 64       # column granularity of a share


def column_slabs(n_cols: int, world: int):
    """[(begin, end)] per rank: contiguous column slabs in whole CHUNK-column chunks, as the engine's parts are cut
    (``part * chunks / G``).  n_cols must be a multiple of CHUNK (the engine pads to it)."""
    if n_cols % CHUNK:
        raise ValueError(f'column count {n_cols} is not a multiple of {CHUNK}')
    chunks = n_cols // CHUNK
    return [(CHUNK * (r * chunks // world), CHUNK * ((r + 1) * chunks // world)) for r in range(world)]


def circle_pair(n: int, r: int, k: int):
    """pair of blocks meeting in round r (0..n-2), slot k (0..n/2-1) of the circle method -- ``circle_pair`` of
    csrc/jacobi_engine.hip"""
    m = n - 1
    if k == 0:
        return (r, m) if r < m else (m, r)
    p, q = (r + k) % m, (r - k) % m
    return (p, q) if p < q else (q, p)


def round_robin_schedule(nb: int):
    """rounds[r] = [(P, Q), ...]: every pair of the nb blocks exactly once per sweep, nb/2 disjoint pairs per round"""
    if nb % 2:
        raise ValueError('number of blocks must be even')
    return [[circle_pair(nb, r, k) for k in range(nb // 2)] for r in range(nb - 1)]


def _jacobi_eigh(G: np.ndarray, sweeps: int = 2) -> np.ndarray:
    """Orthogonal Q that (nearly) diagonalises the symmetric positive semi-definite G: cyclic two-sided Jacobi, the inner
    solver of the engine (rotations from 2 x 2 sub-problems are accurate RELATIVE to the entries they touch, which a
    LAPACK eigh of the Gram matrix is not: with it the iteration stalls at eps (sigma_max / sigma_min)^2)."""
    n = G.shape[0]
    G = np.array(G, copy=True)
    Q = np.eye(n)
    for _ in range(sweeps):
        for r in range(n - 1):
            R = np.eye(n)
            for k in range(n // 2):
                p, q = circle_pair(n, r, k)
                a, d, b = G[p, p], G[q, q], G[p, q]
                if abs(b) <= 1e-300:
                    continue
                delta = d - a
                h = np.hypot(delta, 2.0 * b)
                c2 = 0.5 + 0.5 * abs(delta) / h
                c = np.sqrt(c2)
                sgn = 1.0 if (delta >= 0) == (b >= 0) else -1.0
                sn = sgn * abs(b) / (h * c)
                R[p, p] = R[q, q] = c
                R[p, q], R[q, p] = sn, -sn
            G = R.T @ G @ R
            Q = Q @ R
    return Q[:, np.argsort(-np.diag(G), kind='stable')]


def distributed_block_jacobi(W_slab: np.ndarray, J_slab, group=None, tol: float = 1e-12, max_sweeps: int = 40):
    """One-sided block Jacobi on the rows of W whose columns are distributed over the ranks of `group` (this rank holds
    `W_slab` = all rows x its column slab; `J_slab` likewise for the accumulated transform, or None).  Per round: partial
    Gram matrices of all pairs from the local slab -> ONE all_reduce -> the same eigensolve on every rank -> local update.
    Returns (W_slab, J_slab, sweeps).  Executable specification of the multi-device split (numpy arithmetic)."""
    import torch
    import torch.distributed as dist
    nv = W_slab.shape[0]
    if nv % (2 * JB):
        raise ValueError(f'row count {nv} must be a multiple of {2 * JB} (the engine pads to it)')
    nb = nv // JB
    W = np.array(W_slab, dtype=np.float64, copy=True)
    J = None if J_slab is None else np.array(J_slab, dtype=np.float64, copy=True)
    schedule = round_robin_schedule(nb)
    distributed = dist.is_available() and dist.is_initialized()
    prev_off = 1e300
    for sweep in range(1, max_sweeps + 1):
        off = 0.0
        for pairs in schedule:
            idx = [np.r_[P * JB:(P + 1) * JB, Q * JB:(Q + 1) * JB] for P, Q in pairs]
            grams = np.stack([W[i] @ W[i].T for i in idx])                 # local partials, (pairs, 32, 32)
            if distributed:
                t = torch.from_numpy(grams)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)      # the ONE collective of a round
                grams = t.numpy()
            for i, G in zip(idx, grams):
                d = np.sqrt(np.maximum(np.diag(G), 1e-300))
                C = np.abs(G) / np.outer(d, d)
                np.fill_diagonal(C, 0.0)
                off = max(off, float(C.max()))
                Q = _jacobi_eigh(G)                                         # identical input -> identical Q on every rank
                W[i] = Q.T @ W[i]
                if J is not None:
                    J[i] = Q.T @ J[i]
        if off <= tol or (sweep >= 6 and off <= 64.0 * tol and off >= 0.5 * prev_off):  # converged / at the rounding floor
            return W, J, sweep
        prev_off = off
    return W, J, -1


def split_round_model(world: int, round_us: float = 40.0, mfma_us: float = 19.0, allreduce_us: float = 15.0):
    """Modelled duration (us) of one Jacobi round of the largest block when its columns are split over `world` GPUs:
    the MFMA share of a round (partial Gram + row update, `mfma_us` of `round_us` on one device: DESIGN.md section 4.2)
    shrinks with the column share, the eigensolve / exchange / hand-off part does not, and every round pays one small
    all_reduce over xGMI (`allreduce_us`: RCCL latency of a ~200 KB message, not bandwidth)."""
    if world <= 1:
        return round_us
    return (round_us - mfma_us) + mfma_us / world + allreduce_us
