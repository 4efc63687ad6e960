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
    costs: np.ndarray             # modelled seconds per sector (`svd_cost`: chain length + share of the chip): the shard weights
    layout: PoolLayout            # sector -> rank (LPT by cost); sizes = min(rows, cols)
    s_layout: PoolLayout          # rank-major pool of the singular values (one all_gather makes S global)

    def blocks_of(self, sectors) -> np.ndarray:
        """theta blocks (indices into the contraction plan, ascending = block-table order) of the given sectors"""
        return np.flatnonzero(np.isin(self.sector_of_block, np.asarray(list(sectors), dtype=np.int64)))


# Cost of decomposing ONE sector block inside a batched SVD call, fitted to `scripts/svd_bench.py` on one MI355X (round 3:
# a full-rank 1442^2 block alone 55.5 ms, 1236^2 41 ms, 721^2 17 ms; the rank-deficient 1442^2 block of the theta list 24.4 ms with
# k_eff = 824).  The call is a dependency CHAIN per matrix -- two blocked QRs of k / 32 panel steps (~165 us each) and ~10
# sweeps of k / 16 rounds (~38 us each): ~34 us per unit of k -- plus a throughput share of the nominal flops (the matrices of
# a call run in lockstep and share the chip).  Nominal flops alone (round 2's weight) put a 1442 x 721 block and a 1030^2
# block in the same class although the second one's chain is 1.4 x longer; ranks at N = 4 came out at 28.4 / 24.9 / 24.9 / 17.0 ms.
SVD_COST_PER_K = 3.4e-5          # seconds per unit of k = min(m, n): the chain
SVD_COST_PER_FLOP = 1.8e-13      # seconds per nominal flop (4 m n^2 + 8 n^3): the share of the chip


def svd_cost(m: int, n: int) -> float:
    """modelled seconds of one sector block's SVD inside a batched call (the LPT weight of `theta_sector_plan`)"""
    k, big = min(m, n), max(m, n)
    return SVD_COST_PER_K * k + SVD_COST_PER_FLOP * (4.0 * big * k * k + 8.0 * k ** 3)


def theta_sector_plan(plan, a, num_codomain: int, world: int) -> SectorPlan:
    """Group the result blocks of a contraction plan (``abelian.ComposePlan``) by the coupled charge of their first
    `num_codomain` legs -- the sectors ``combine_legs_to_matrix`` produces, in its order -- and LPT-assign the sectors to
    `world` ranks by the fitted cost model `svd_cost`.  Pure int64 host work."""
    from . import abelian as ab
    cache = plan.__dict__.setdefault('_sector_plans', {})     # (a plan is a function of legs and block tables only, and is itself cached)
    hit = cache.get((num_codomain, world))
    if hit is not None:
        return hit
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
    costs = np.array([svd_cost(*s) for s in shapes])
    ks = np.array([min(s) for s in shapes], dtype=np.int64)
    layout = make_layout(ks, costs, world)
    s_layout = layout_for_owner(ks, layout.owner, world)
    cache[(num_codomain, world)] = out = SectorPlan(sector_of_block, shapes, costs, layout, s_layout)
    return out
