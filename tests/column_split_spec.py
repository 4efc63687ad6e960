"""Splitting ONE block's Jacobi iteration across GPUs by column slabs: an executable specification and its cost model
(DESIGN.md section 5).  Test infrastructure and documentation, not product code: it runs on the CPU over `gloo`
(tests/test_distributed.py), its 32 x 32 solves are a numpy two-sided Jacobi, nothing in `cyten_amd/` imports it.  The
split was modelled and NOT built on the device -- every N > 1 makes a round longer (`split_round_model`)."""
import numpy as np

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
