#!/usr/bin/env python3
"""bench.py -- block-sparse tdot + SVD throughput of the MI355X-native cyten block backend.

Metric (BASELINE.json): "block-sparse tdot+SVD GFLOP/s (fp64) at chi=4096 U(1) MPS, 1/2/4/8 GPUs".

One *step* = one pass of the hot path over one synthetic two-site theta (inputs resident in HBM):
  1. host sector matching + grouped-GEMM launch      theta = A . B          (cyten.tdot)
  2. combine legs to one matrix per coupled charge   (zero fill + batched strided scatter)
  3. batched block-Jacobi SVD of all sector blocks   (cyten.svd)
  4. truncation: selection on the device, batched gather of the kept U / S / Vh  (cyten.truncated_svd)
`value` = (sum 2MNK over matched pairs + sum (4 m n^2 + 8 n^3) over SVD blocks) / step time, the
algorithmic counts of SURVEY.md section 8d -- independent of the flops the Jacobi iteration
really executes.

N > 1 (one process per GPU): strong scaling of the SAME theta.  `python bench.py --gpus N` without a
torch.distributed environment starts the N ranks itself (python -m torch.distributed.run, before anything touches the
GPU in this process); launched by the driver as `python -m torch.distributed.run ... bench.py --gpus N` it reads
RANK / LOCAL_RANK / WORLD_SIZE.  The coupled-charge sectors are LPT-sharded over the ranks by nominal SVD flops; a rank
contracts, combines and decomposes ONLY the theta blocks of its own sectors (theta never leaves the rank), one small
all_gather makes every singular value known everywhere (the truncation is a global selection), and one all_gather of the
KEPT U / S / Vh columns completes the truncated factors on every rank (cyten_amd.sharding).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F64_SPEC_TFLOPS = 78.6   # 256 CU x 4 SIMD x 32 FLOP/clk/SIMD x 2.4 GHz (AMD datasheet FP64 matrix)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--chi', type=int, default=4096)
    ap.add_argument('--symmetry', choices=['u1', 'u1u1'], default='u1')
    ap.add_argument('--chi-max', type=int, default=None, help='truncation target (default: chi)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=100.0, help='seconds of host time the CPU baseline may use')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the untimed extras after the timed loop (U(1)xU(1) GEMM roofline, torch.linalg.svd on the device)')
    ap.add_argument('--master-port', type=int, default=29533)
    ap.add_argument('--dry-run', action='store_true', help='print the launch decision as JSON and exit (no GPU)')
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------- launcher (no GPU calls here)

def launcher_command(args, argv, env) -> list | None:
    """The command that starts the N ranks, or None when this process IS a rank (or N == 1).

    `python bench.py --gpus N` with N > 1 and no torch.distributed environment (RANK / WORLD_SIZE unset) re-launches
    itself under `python -m torch.distributed.run`, one process per GPU, rendezvous on 127.0.0.1."""
    if args.gpus <= 1 or 'WORLD_SIZE' in env or 'RANK' in env:
        return None
    rest = [a for a in argv if a != '--dry-run']
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(args.master_port), os.path.abspath(__file__), *rest]


# ------------------------------------------------------------------------------------------------------ the hot path

_HipBlock = _c_strides = None


def pool_view(bb, pool, offset, shape):
    global _HipBlock, _c_strides
    if _HipBlock is None:  # (imported lazily: bench.py must parse its arguments without touching the GPU)
        from cyten_amd.block_backend import HipBlock as _HipBlock, _c_strides
    shape = tuple(map(int, shape))
    return _HipBlock._trusted(bb, pool, int(offset), shape, _c_strides(shape), True)   # (carved out of a pool: contiguous)


class Timer:
    """HIP events on the context's launch stream around one phase of every timed step."""

    def __init__(self, ctx, n=64):
        self.ctx = ctx
        self.ev = [(ctx.event(), ctx.event()) for _ in range(n)]
        self.k = 0

    def __enter__(self):
        self.on = self.k < len(self.ev)
        if self.on:
            self.ctx.record(self.ev[self.k][0])
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.record(self.ev[self.k][1])
            self.k += 1

    def arm_kernel(self):
        """time the KERNEL of the next asynchronous grouped-GEMM launch: the library records the pair directly around it, behind
        the descriptor upload (cyb_ctx_time_next_gemm) -- what rocprofv3 --kernel-trace reports for that kernel"""
        if self.k < len(self.ev):
            self.ctx.time_next_gemm(*self.ev[self.k])
            self.k += 1

    def ms(self):
        return [self.ctx.elapsed_ms(a, b) for a, b in self.ev[:self.k]]


class _Off:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass


class ThetaStep:
    """The hot path of one bond, sharded over `world` ranks (world = 1: plain single-GPU path)."""

    def __init__(self, bb, A, B, chi_max, rank=0, world=1):
        from cyten_amd import abelian as ab
        from cyten_amd import sharding
        self.bb, self.ab, self.sharding = bb, ab, sharding
        self.rank, self.world, self.chi_max = rank, world, chi_max
        self.a, self.b = ab.AbelianTensor.from_spec(bb, A), ab.AbelianTensor.from_spec(bb, B)
        self.t_gemm, self.t_svd, self.t_coll = Timer(bb.ctx), Timer(bb.ctx), Timer(bb.ctx, 128)
        self.last = None

    def step(self, timed=True, lazy_null=False):
        bb, ab, sh = self.bb, self.ab, self.sharding
        a, b = self.a, self.b
        # ---- 1. tdot: host sector matching; the sectors (coupled charges of the combined matrix) are the shard units
        plan = ab.compose_plan(a, b, 1)
        sp = sh.theta_sector_plan(plan, a, 2, self.world)          # which theta block feeds which sector, sector -> rank
        mine_sec = sp.layout.local_units(self.rank)
        mine_blk = sp.blocks_of(mine_sec)                            # theta blocks this rank computes, block-table order
        shp = np.array(plan.res_shapes, dtype=np.int64).reshape(len(plan.res_shapes), 4)
        sizes = shp.prod(axis=1)
        lay = sh.make_layout(sizes[mine_blk], np.ones(len(mine_blk)), 1)
        pool = bb.ctx.empty(lay.total)
        pa, pb = a.block_ptrs(), b.block_ptrs()
        if plan.native is not None and pa is not None and pb is not None:
            # the contraction behind the C-ABI: descriptors are built inside the library from the plan and the address tables
            # (cyb_compose_plan_enqueue_f64), results land in this rank's pool, blocks become objects only if somebody looks
            out_ptrs = pool.data_ptr() + 8 * np.asarray(lay.offset, dtype=np.int64)
            if timed:
                self.t_gemm.arm_kernel()
            self.gemm_flops_local, self.gemm_bytes_local = ab.compose_enqueue(bb, plan, pa, pb, out_ptrs, which=mine_blk)
            theta = ab.AbelianTensor(a.symmetry, plan.legs, ab.LazyBlocks(bb, pool, lay.offset, [plan.res_shapes[u] for u in mine_blk]),
                                     plan.res_block_inds[mine_blk], 2, ptrs=out_ptrs)
            gemm = None
        else:
            a2, b2 = ab._compose_operands(bb, a, b, 1, plan)
            groups = [[(a2[i], b2[j]) for i, j in plan.pairs[u]] for u in mine_blk]
            outs = [pool_view(bb, pool, lay.offset[k], (plan.res_shapes[u][0] * plan.res_shapes[u][1],
                                                        plan.res_shapes[u][2] * plan.res_shapes[u][3]))
                    for k, u in enumerate(mine_blk)]
            gemm = bb.make_gemm_plan(groups, outs)
            with (self.t_gemm if timed else _Off()):
                gemm.run()
            self.gemm_flops_local, self.gemm_bytes_local = gemm.flops, gemm.bytes
            theta_blocks = [pool_view(bb, pool, lay.offset[k], plan.res_shapes[u]) for k, u in enumerate(mine_blk)]
            theta = ab.AbelianTensor(a.symmetry, plan.legs, theta_blocks, plan.res_block_inds[mine_blk], 2)
        # ---- 2./3. combine this rank's sectors to matrices, batched SVD into the rank's segment of the factor pool
        mv = ab.combine_legs_to_matrix(bb, theta, 2)
        assert len(mv.blocks) == len(mine_sec)
        shapes_all = sp.shapes
        k_all = np.array([min(s) for s in shapes_all], dtype=np.int64)
        s_lay = sp.s_layout
        s_pool = bb.ctx.empty(s_lay.total)
        fac = [bb.ctx.empty(int(m * min(m, n) + min(m, n) * n)) for (m, n) in (shapes_all[u] for u in mine_sec)]

        def usv_views(k, u):
            m, n = shapes_all[u]
            kk = min(m, n)
            return (pool_view(bb, fac[k], 0, (m, kk)), pool_view(bb, s_pool, s_lay.offset[u], (kk,)),
                    pool_view(bb, fac[k], m * kk, (kk, n)))

        usv = [usv_views(k, u) for k, u in enumerate(mine_sec)]
        ranks = None
        with (self.t_svd if timed else _Off()):
            if lazy_null:   # the truncating caller's form (cyb_svd_batched_ex_f64): null vectors only if they will be kept
                _, ranks = bb.matrix_svd_batched(list(mv.blocks), outs=usv, null_vectors=False, return_rank=True)
            else:
                bb.matrix_svd_batched(list(mv.blocks), outs=usv)
        # ---- 4. truncation: every singular value everywhere (one small collective), the same selection on every rank
        with (self.t_coll if timed else _Off()):
            sh.allgather_pool(s_pool, s_lay, self.rank)
        S = [pool_view(bb, s_pool, s_lay.offset[u], (int(k_all[u]),)) for u in range(len(k_all))]
        if sum(s.size for s in S) <= bb.TRUNCATE_MAX:  # selection on the device: the host reads counts, err, new_norm
            masks, _, err, new_norm = bb.truncate_select(S, chi_max=self.chi_max)
        else:
            masks, err, new_norm = ab.truncate_singular_values(bb, S, chi_max=self.chi_max)
        kept_n = np.array([int(m.n if hasattr(m, 'n') else m.sum()) for m in masks], dtype=np.int64)
        if ranks is not None:   # a sector that keeps a numerically zero singular value needs its null vectors after all
            redo = [k for k, u in enumerate(mine_sec) if kept_n[u] > ranks[k]]
            if redo:
                bb.matrix_svd_batched([mv.blocks[k] for k in redo], outs=[usv[k] for k in redo])
        # kept columns of this rank's U / Vh go straight into its segment of the kept-factor pool; ONE all_gather
        # afterwards leaves the truncated factors of every sector addressable on every rank
        ksz = np.array([shapes_all[u][0] * kept_n[u] + kept_n[u] + kept_n[u] * shapes_all[u][1] for u in range(len(k_all))])
        k_lay = sh.layout_for_owner(ksz, sp.layout.owner, self.world)
        k_pool = bb.ctx.empty(k_lay.total)

        def kept_views(u):
            m, n = shapes_all[u]
            c, o = int(kept_n[u]), int(k_lay.offset[u])
            return (pool_view(bb, k_pool, o, (m, c)), pool_view(bb, k_pool, o + m * c, (c,)),
                    pool_view(bb, k_pool, o + m * c + c, (c, n)))

        jobs, outs_k = [], []
        for k, u in enumerate(mine_sec):
            uo, so, vo = kept_views(u)
            jobs += [(usv[k][0], masks[u], 1), (usv[k][1], masks[u], 0), (usv[k][2], masks[u], 0)]
            outs_k += [uo, so, vo]
        bb.mask_gather_many(jobs, outs=outs_k)
        with (self.t_coll if timed else _Off()):
            sh.allgather_pool(k_pool, k_lay, self.rank)
        kept = [kept_views(u) for u in range(len(k_all))]
        if gemm is not None:
            gemm.destroy()
        svd_cost = sp.costs
        self.last = dict(theta=theta, mv=mv, usv=usv, masks=masks, err=err, new_norm=new_norm, kept=kept,
                         shapes=shapes_all, local_sectors=mine_sec, plan=plan, kept_n=kept_n,
                         imbalance_svd=sp.layout.imbalance(svd_cost))
        return self.last


# ------------------------------------------------------------------------------------------------------ CPU baseline

def _blas_info():
    """BLAS vendor / version / threading of the numpy + scipy wheels on this host (threadpoolctl; numpy.show_config as
    the fallback), and the largest thread count any pool reports."""
    out = {'pools': [], 'max_threads': os.cpu_count() or 1}
    try:
        from threadpoolctl import threadpool_info
        infos = threadpool_info()
        out['pools'] = [{k: p.get(k) for k in ('user_api', 'internal_api', 'version', 'threading_layer', 'architecture',
                                               'num_threads', 'prefix')} for p in infos]
        out['max_threads'] = int(max([p.get('num_threads', 1) for p in infos] + [1]))
    except Exception:
        pass
    try:
        cfg = np.show_config(mode='dicts')
        blas = cfg.get('Build Dependencies', {}).get('blas', {})
        out['numpy_blas'] = {k: blas.get(k) for k in ('name', 'version', 'openblas configuration') if k in blas}
    except Exception:
        pass
    return out


def cpu_baseline(A, B, chi_max, budget_s):
    """The oracle (CPU restatement calling the same numpy/scipy routines as the reference's NumpyBlockBackend:
    np.dot per matched pair numpy.cpp:1218-1225, scipy.linalg.svd per block numpy.cpp:1247-1297, host truncation
    tensor_backend.cpp:139-242) timed on this host's cores over a sweep of BLAS thread counts.

    The full chi-sized step is run at every thread count that fits the time budget: a warm call, then the median of
    up to 5 repetitions; tdot and SVD seconds are reported separately for the best count.  The single-thread point is
    measured on a bounded sample (all GEMMs + every third SVD block, scaled by the nominal-flop ratio) when a full
    single-thread step would not fit."""
    from oracle import abelian_ref as ref
    from cyten_amd import workloads as wl
    try:
        from threadpoolctl import threadpool_limits
    except Exception:   # pragma: no cover - threadpoolctl is in the image; without it: one measurement at the default
        threadpool_limits = None
    info = _blas_info()
    all_t = int(info['max_threads'])
    t_start = time.perf_counter()

    def phases():
        t0 = time.perf_counter()
        blocks, bi, _ = ref.compose(A, B, 1)
        t1 = time.perf_counter()

        class _T:
            pass
        th = _T()
        th.moduli, th.legs = A.moduli, list(A.legs[:-1]) + list(B.legs[1:])
        th.block_inds, th.blocks = bi, blocks
        _, mats, _, _ = ref.combine_to_matrix(th, len(A.legs) - 1)
        t2 = time.perf_counter()
        usv = ref.svd_blocks(mats)
        t3 = time.perf_counter()
        S_all = np.concatenate([s for _, s, _ in usv])
        ref.truncation_selection(S_all, chi_max=chi_max)
        t4 = time.perf_counter()
        return dict(total=t4 - t0, tdot=t1 - t0, combine=t2 - t1, svd=t3 - t2, truncation=t4 - t3), mats

    mats_keep = [None]

    def run_at(nt, max_reps):
        ctxm = threadpool_limits(limits=nt) if threadpool_limits else _Off()
        with ctxm:
            first, mats_keep[0] = phases()           # warm call (discarded)
            reps = []
            for _ in range(max_reps):
                reps.append(phases()[0])
                spent = time.perf_counter() - t_start
                if spent + 2 * first['total'] > budget_s and len(reps) >= 3:
                    break
        med = sorted(reps, key=lambda r: r['total'])[len(reps) // 2]
        return med, len(reps)

    counts = sorted({t for t in (8, 16, 32, all_t) if t <= all_t}) if threadpool_limits else [all_t]
    sweep, per = {}, {}
    for nt in counts:
        if time.perf_counter() - t_start > 0.8 * budget_s and sweep:
            break
        med, n = run_at(nt, 5)
        sweep[nt] = round(med['total'], 4)
        per[nt] = (med, n)
    best = min(sweep, key=sweep.get)
    med_best, n_best = per[best]
    # single thread: bounded sample
    one = None
    if threadpool_limits and time.perf_counter() - t_start < budget_s:
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            ref.compose(A, B, 1)
            t_dot = time.perf_counter() - t0
            mats = mats_keep[0]
            sample = list(range(0, len(mats), 3))
            t0 = time.perf_counter()
            ref.svd_blocks([mats[i] for i in sample])
            t_s = time.perf_counter() - t0
        nom_all = wl.svd_nominal_flops([m.shape for m in mats])
        nom_smp = wl.svd_nominal_flops([mats[i].shape for i in sample])
        one = {'seconds_per_step_estimate': round(t_dot + t_s * nom_all / nom_smp, 3), 'tdot_s': round(t_dot, 3),
               'svd_sample_s': round(t_s, 3),
               'sample': f'all GEMMs + SVD of every third sector block ({len(sample)} of {len(mats)}, '
                         f'{nom_smp / nom_all:.2f} of the nominal SVD flops), one call, scaled by the nominal-flop ratio'}
    return dict(sweep_seconds_per_step=sweep, best_threads=best, all_threads=all_t, seconds_best=sweep[best],
                seconds_all=sweep.get(all_t), reps_best=n_best, split_best={k: round(v, 4) for k, v in med_best.items()},
                single_thread=one, blas=info, host_seconds_used=round(time.perf_counter() - t_start, 1))


# ------------------------------------------------------------------------------------------- untimed extras (rank 0)

def _sha16(path):
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(kind, key):
    """Counter-measured HBM bytes per launch from the committed rocprofv3 --pmc summary of THIS round -- only if the summary
    was taken on the kernel source that is being benchmarked (the summary records the source hash); otherwise None."""
    src = {'gemm': ['gemm_grouped.hip'], 'svd': ['jacobi_engine.hip', 'svd_jacobi.hip', 'blocked_qr.hip']}[kind]
    now = {s: _sha16(os.path.join(ROOT, 'cyten_amd', 'csrc', s)) for s in src}
    why = 'no PMC summary committed yet'
    for rnd in ('r03', 'r02'):          # newest round first; a summary counts only if it was taken on the source being benchmarked
        name = f'{rnd}_{kind}_pmc_summary.json'
        path = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            js = json.load(f)
        if js.get('source_sha16') != now:
            why = f'profiles/{name} is stale (taken on {js.get("source_sha16")}, source now {now}): not quoted'
            continue
        ent = js.get(key)
        if not ent or 'hbm_bytes_corrected' not in ent:
            why = f'no entry {key} in profiles/{name}'
            continue
        return ent['hbm_bytes_corrected'], f'profiles/{name} ({key}), separate --pmc passes, FETCH_SIZE x2 + WRITE_SIZE'
    return None, why


def u1u1_gemm_roofline(bb, reps=10):
    """The north-star list (U(1)xU(1) chi=4096 theta: 728 GEMMs) through the same grouped launch, HIP events."""
    from cyten_amd import abelian as ab
    from cyten_amd import workloads as wl
    A, B = wl.config_u1u1_mps(4096)
    a, b = ab.AbelianTensor.from_spec(bb, A), ab.AbelianTensor.from_spec(bb, B)
    plan = ab.compose_plan(a, b, 1)
    a2, b2 = ab._compose_operands(bb, a, b, 1, plan)
    groups = [[(a2[i], b2[j]) for i, j in g] for g in plan.pairs]
    sizes = [int(np.prod(s)) for s in plan.res_shapes]
    pool = bb.ctx.empty(sum(sizes))
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    outs = [pool_view(bb, pool, o, (s[0] * s[1], s[2] * s[3])) for o, s in zip(offs, plan.res_shapes)]
    gemm = bb.make_gemm_plan(groups, outs)
    for _ in range(3):
        gemm.run()
    t = Timer(bb.ctx, reps)
    for _ in range(reps):
        with t:
            gemm.run()
    bb.synchronize()
    ms = float(np.mean(t.ms()))
    ach = gemm.flops / (ms * 1e-3) / 1e12
    traffic, note = pmc_traffic('gemm', 'theta_chi4096_u1u1')
    out = {'kernel': 'gemm_grouped_kernel', 'workload': f'U(1)xU(1) chi=4096 theta: {len(groups)} result blocks, '
           f'{sum(len(g) for g in groups)} GEMMs', 'bound': 'mfma', 'achieved': round(ach, 3), 'peak': MFMA_F64_SPEC_TFLOPS,
           'unit': 'TFLOP/s', 'frac': round(ach / MFMA_F64_SPEC_TFLOPS, 4), 'traffic': traffic, 'traffic_source': note,
           'flops_per_launch': gemm.flops, 'algorithmic_bytes_per_launch': gemm.bytes, 'avg_launch_ms': round(ms, 4),
           'launches_timed': reps}
    gemm.destroy()
    return out


def dominant_block_roofline(bb, reps=20):
    """north_star's literal wording: the DOMINANT-BLOCK GEMM alone -- 824 x 721 x 824 (the largest product of the U(1) chi=4096
    theta) and 474^3 (the largest of the U(1)xU(1) list) -- through the same grouped launch, one problem per launch."""
    rng = np.random.default_rng(1)
    out = {}
    for name, (M, K, N) in (('u1_824x721x824', (824, 721, 824)), ('u1u1_474x474x474', (474, 474, 474))):
        A, B = bb.as_block(rng.standard_normal((M, K))), bb.as_block(rng.standard_normal((K, N)))
        C = bb.empty_block((M, N))
        gemm = bb.make_gemm_plan([[(A, B)]], [C])
        for _ in range(3):
            gemm.run()
        t = Timer(bb.ctx, reps)
        for _ in range(reps):
            with t:
                gemm.run()
        bb.synchronize()
        ms = float(np.mean(t.ms()))
        ach = gemm.flops / (ms * 1e-3) / 1e12
        out[name] = {'bound': 'mfma', 'achieved': round(ach, 3), 'peak': MFMA_F64_SPEC_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(ach / MFMA_F64_SPEC_TFLOPS, 4), 'flops_per_launch': gemm.flops,
                     'algorithmic_bytes_per_launch': gemm.bytes, 'avg_launch_ms': round(ms, 4), 'launches_timed': reps}
        gemm.destroy()
    out['note'] = ('one GEMM per launch: a single 824 x 824 result is 49 tiles of 128 x 128 for 512 workgroup slots, so the launch is '
                   'occupancy- and launch-latency bound; the block LIST (roofline, roofline_u1u1) is what a tdot issues')
    return out


def fullrank_svd_roofline(bb, shapes, cpu_threads, budget_s=25.0, reps=3):
    """The SVD list of the headline theta with FULL-RANK blocks: the same 15 shapes filled with standard normal entries
    (the benchmark's theta = A.B has half the rank of its extents in every sector, which the early-stopping QR and the free
    null-space completion exploit; a converged DMRG theta is full rank).  HIP-event time of `cyb_svd_batched_f64` and the
    scipy loop of the oracle on the host beside it."""
    from oracle import block_ops as ops
    from cyten_amd import workloads as wl
    rng = np.random.default_rng(7)
    mats = [rng.standard_normal(tuple(int(x) for x in s)) for s in shapes]
    blocks = [bb.as_block(m) for m in mats]
    _, info = bb.matrix_svd_batched(blocks, return_info=True)
    t = Timer(bb.ctx, reps)
    for _ in range(reps):
        with t:
            res = bb.matrix_svd_batched(blocks)
    bb.synchronize()
    ms = float(np.mean(t.ms()))
    flops = wl.svd_nominal_flops([m.shape for m in mats])
    ach = flops / (ms * 1e-3) / 1e12
    worst = 0.0
    i_big = int(np.argmax([m.size for m in mats]))
    u, s_, vh = (bb.to_numpy(x) for x in res[i_big])
    worst = float(np.abs((u * s_) @ vh - mats[i_big]).max() / np.linalg.norm(mats[i_big]))
    out = {'kernel': 'cyb_svd_batched_f64', 'workload': f'the {len(mats)} block shapes of the headline list, full-rank Gaussian entries',
           'bound': 'mfma', 'achieved': round(ach, 4), 'peak': MFMA_F64_SPEC_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / MFMA_F64_SPEC_TFLOPS, 5),
           'nominal_flops_per_call': flops, 'avg_call_ms': round(ms, 3), 'sweeps': [int(x) for x in info],
           'reconstruction_error_largest_block': worst}
    try:
        from threadpoolctl import threadpool_limits
        ctxm = threadpool_limits(limits=cpu_threads)
    except Exception:
        ctxm = _Off()
    with ctxm:
        t0 = time.perf_counter()
        for m in mats:
            ops.matrix_svd(m)
        first = time.perf_counter() - t0
        ts = []
        while len(ts) < 3 and (time.perf_counter() - t0) + first < budget_s:
            t1 = time.perf_counter()
            for m in mats:
                ops.matrix_svd(m)
            ts.append(time.perf_counter() - t1)
    cpu_s = float(np.median(ts)) if ts else first
    out['cpu_baseline'] = {'seconds_per_list': round(cpu_s, 4), 'cores': cpu_threads, 'kind': 'port',
                           'sample': f'scipy.linalg.svd per block over the same list, {len(ts) or 1} call(s) after one warm call',
                           'speedup': round(cpu_s / (ms * 1e-3), 2)}
    return out


def torch_svd_same_hw(bb, mats, budget_s=60.0):
    """torch.linalg.svd on the same device over the same block list, one call per block -- the reference's EXISTING GPU
    route (TorchBlockBackend::matrix_svd -> torch::linalg_svd, /root/reference/src/block_backend/torch.cpp:1360-1393).
    A comparison number only: nothing in the product calls torch.linalg."""
    import torch
    ts = []
    t_start = time.perf_counter()
    tens = [torch.from_numpy(bb.to_numpy(m)).to(bb.ctx.device) for m in mats]
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in tens:
            torch.linalg.svd(t, full_matrices=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    return {'routine': 'torch.linalg.svd(full_matrices=False), default driver, one call per block (torch ' + torch.__version__ + ')',
            'seconds_per_list': round(min(ts[1:] or ts), 4), 'first_call_seconds': round(ts[0], 4), 'calls': len(ts)}


# --------------------------------------------------------------------------------------------------------------- main

def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    cmd = launcher_command(args, argv, os.environ)
    if args.dry_run:
        print(json.dumps({'launch': cmd, 'n_gpus': args.gpus,
                          'role': 'launcher' if cmd else ('rank' if 'RANK' in os.environ else 'single process')}))
        return 0
    if cmd is not None:      # start the ranks BEFORE anything in this process touches the GPU; exit with their code
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        return subprocess.run(cmd, env=env).returncode

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # BENCH_FORCE_DIST=1 exercises the RCCL code path (init, in-place all_gather, barrier) with one rank
    if world > 1 or os.environ.get('BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(args.master_port))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        rccl_ranks = dist.get_world_size()
    else:
        dist = None
        rccl_ranks = 0
        torch.cuda.set_device(0)
    if world != args.gpus and rank == 0:
        print(f'[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE', file=sys.stderr)

    from cyten_amd import workloads as wl
    from cyten_amd.block_backend import HipBlockBackend

    bb = HipBlockBackend(f'cuda:{local_rank}')
    chi_max = args.chi_max or args.chi
    if args.symmetry == 'u1':
        A, B = wl.config_u1_mps(args.chi)
        workload = f'U(1) MPS two-site theta tdot + truncated SVD, chi={args.chi}, fp64'
    else:
        A, B = wl.config_u1u1_mps(args.chi)
        workload = f'U(1)xU(1) MPS two-site theta tdot + truncated SVD, chi={args.chi}, fp64'
    gemm_flops, gemm_bytes, n_theta_blocks = wl.theta_flops(A, B)

    runner = ThetaStep(bb, A, B, chi_max, rank, world)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.step(timed=False)
    # Host runtime setting, not skipped work: the objects alive after the warm-up (torch, the library, the operands) are
    # moved out of the cyclic collector's working set.  A step creates ~10^4 short-lived view objects on the 728-block
    # U(1)xU(1) list, and every full collection they trigger would otherwise traverse everything alive: 41.8 -> 33.8 ms per
    # step there, nothing on the 54-block headline list (measured).
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = runner.step(timed=True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    svd_flops = wl.svd_nominal_flops(res['shapes'])
    total_flops = gemm_flops + svd_flops
    ms_per_step = 1e3 * dt / args.steps
    value = total_flops / (dt / args.steps) / 1e9

    # roofline of the dominant north-star kernel: the grouped fp64 MFMA GEMM, HIP events on the launch stream
    gms = runner.t_gemm.ms()
    gemm_ms = float(np.mean(gms)) if gms else float('nan')
    achieved = runner.gemm_flops_local / (gemm_ms * 1e-3) / 1e12 if gms else float('nan')
    headline = world == 1 and args.chi == 4096 and args.symmetry == 'u1'
    traffic, tnote = pmc_traffic('gemm', 'theta_chi4096_u1') if headline else (None, 'not the headline configuration')
    roofline = {
        'kernel': 'gemm_grouped_kernel (one persistent launch, all tile classes; the 128x128 f64-MFMA tile carries >99% of the flops)',
        'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': MFMA_F64_SPEC_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / MFMA_F64_SPEC_TFLOPS, 4), 'traffic': traffic, 'traffic_source': tnote,
        'flops_per_launch': runner.gemm_flops_local, 'algorithmic_bytes_per_launch': runner.gemm_bytes_local,
        'avg_launch_ms': round(gemm_ms, 4),
        'reference_same_hw': 'rocBLAS dgemm 4096^3 = 72 TFLOP/s (0.92 of peak); this kernel 62 TFLOP/s (0.79) on the same uniform GEMM',
        'share_of_step': round(gemm_ms / ms_per_step, 4) if gms else None,
        'note': 'the kernel north_star names (dominant-block GEMM of the tdot); it is about one per cent of the step -- the step IS the '
                'batched SVD (roofline_svd), whose nominal-flop rate is a latency chain, not a throughput kernel',
    }
    if rank == 0 and world == 1 and not args.no_extras:
        try:    # the measured ceiling beside the spec one: back-to-back v_mfma_f64_16x16x4_f64 on every SIMD, outside the timed region
            tf, _ = bb.ctx.mfma_f64_peak(200000, 4, 4)
            roofline['peak_measured'] = {'value': round(tf, 2), 'unit': 'TFLOP/s', 'frac_of_measured': round(achieved / tf, 4),
                                         'how': 'cyb_mfma_f64_peak: 4 waves/SIMD x 4 independent accumulators in AGPRs, issue loop only'}
        except Exception as exc:   # (measurement extra: never fails the bench line)
            roofline['peak_measured'] = {'error': repr(exc)}
    # the phase that IS the step: the batched SVD (QR preconditioning + block Jacobi + completion), nominal flops
    sms = runner.t_svd.ms()
    svd_ms = float(np.mean(sms)) if sms else float('nan')
    local_shapes = [res['shapes'][u] for u in res['local_sectors']]
    svd_flops_local = wl.svd_nominal_flops(local_shapes)
    svd_bytes_local = float(sum(8 * (m * n + m * min(m, n) + min(m, n) + min(m, n) * n) for m, n in local_shapes))
    s_ach = svd_flops_local / (svd_ms * 1e-3) / 1e12 if sms else float('nan')
    straffic, snote = pmc_traffic('svd', 'theta_chi4096_u1') if headline else (None, 'not the headline configuration')
    roofline_svd = {
        'kernel': 'cyb_svd_batched_f64 (blocked Householder QR -> block one-sided Jacobi -> completion), all kernels of the call',
        'bound': 'mfma', 'achieved': round(s_ach, 4), 'peak': MFMA_F64_SPEC_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(s_ach / MFMA_F64_SPEC_TFLOPS, 5), 'traffic': straffic, 'traffic_source': snote,
        'nominal_flops_per_call': svd_flops_local, 'algorithmic_bytes_per_call': svd_bytes_local,
        'avg_call_ms': round(svd_ms, 3), 'blocks': len(local_shapes), 'share_of_step': round(svd_ms / ms_per_step, 4) if sms else None,
        'note': 'nominal 4mn^2+8n^3 (SURVEY 8d) over HIP-event time of the whole batched call; the Jacobi iteration is a '
                'latency chain of dependent rounds, not a throughput kernel (DESIGN.md 4.2)',
    }

    # per-rank phase times (HIP events on each rank's launch stream), gathered on rank 0: where a sharded step spends its time
    cms = runner.t_coll.ms()
    mine = [gemm_ms if gms else 0.0, svd_ms if sms else 0.0, 2.0 * float(np.mean(cms)) if cms else 0.0, float(len(local_shapes)),
            float(max((min(s) for s in local_shapes), default=0))]
    per_rank = [mine]
    if dist is not None:
        t = torch.tensor(mine, dtype=torch.float64, device='cuda')
        allt = torch.empty(world * len(mine), dtype=torch.float64, device='cuda')
        dist.all_gather_into_tensor(allt, t)
        per_rank = allt.cpu().numpy().reshape(world, len(mine)).tolist()
    ranks_report = [{'rank': r, 'gemm_ms': round(x[0], 4), 'svd_ms': round(x[1], 3), 'collectives_ms': round(x[2], 4), 'sectors': int(x[3]),
                     'largest_k': int(x[4])} for r, x in enumerate(per_rank)]

    out = {
        'metric': 'block-sparse tdot+SVD GFLOP/s (fp64)', 'value': round(value, 2), 'unit': 'GFLOP/s',
        'n_gpus': world, 'rccl_ranks': rccl_ranks, 'steps': args.steps, 'warmup': args.warmup, 'host_gc': 'gc.freeze() after the warm-up steps',
        'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': workload, 'chi': args.chi, 'chi_max': chi_max, 'theta_gemms': len(res['plan'].pairs),
                   'gemm_gflop': round(gemm_flops / 1e9, 3), 'svd_blocks': len(res['shapes']),
                   'svd_nominal_gflop': round(svd_flops / 1e9, 3),
                   'largest_svd_block': list(max(res['shapes'], key=lambda s: s[0] * s[1])),
                   'parallelism': f'coupled-charge sectors sharded x{world} (theta stays on its rank; all_gather of S and of the '
                                  f'kept U/S/Vh)' if world > 1 else 'single GPU',
                   'shard_imbalance': {'svd_cost_model': round(res['imbalance_svd'], 3)}},
        'roofline': roofline,
        'roofline_svd': roofline_svd,
        'truncation': {'err': res['err'], 'new_norm': res['new_norm'], 'kept': int(res['kept_n'].sum())},
        'ranks': ranks_report,
    }
    if rank == 0 and world == 1 and not args.no_extras:
        try:    # the same step as a truncating caller runs it (not the metric: the null vectors of the rank-deficient
            # sectors are never formed, see cyb_svd_batched_ex_f64); same truncation result is asserted
            for _ in range(2):
                lz = runner.step(timed=False, lazy_null=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                lz = runner.step(timed=False, lazy_null=True)
            torch.cuda.synchronize()
            lazy_ms = 1e3 * (time.perf_counter() - t1) / 3
            same = abs(lz['err'] - res['err']) <= 1e-10 * (res['err'] + res['new_norm']) and int(lz['kept_n'].sum()) == int(res['kept_n'].sum())
            out['truncating_caller'] = {'ms_per_step': round(lazy_ms, 3), 'same_truncation': bool(same),
                                        'note': 'tdot + truncated SVD with CYB_SVD_SKIP_NULL_VECTORS: the singular vectors of numerically '
                                                'zero singular values, which the truncation discards, are not completed; NOT the metric'}
        except Exception as e:
            out['truncating_caller'] = {'error': repr(e)}
        try:
            out['roofline_u1u1'] = u1u1_gemm_roofline(bb)
        except Exception as e:  # an extra must never take the headline line down
            out['roofline_u1u1'] = {'error': repr(e)}
        try:
            out['roofline_dominant_block'] = dominant_block_roofline(bb)
        except Exception as e:
            out['roofline_dominant_block'] = {'error': repr(e)}
        try:
            ref_hw = torch_svd_same_hw(bb, list(res['mv'].blocks))
            ref_hw['this_backend_seconds_per_list'] = round(svd_ms * 1e-3, 4)
            ref_hw['speedup'] = round(ref_hw['seconds_per_list'] / (svd_ms * 1e-3), 2)
            out['roofline_svd']['reference_same_hw'] = ref_hw
        except Exception as e:
            out['roofline_svd']['reference_same_hw'] = {'error': repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(A, B, chi_max, args.cpu_budget)
        best_s = cb['seconds_best']
        out['cpu_baseline'] = {'value': round(total_flops / best_s / 1e9, 2), 'unit': 'GFLOP/s', 'cores': cb['best_threads'],
                               'kind': 'port', 'seconds_per_step': best_s,
                               'sample': f'the full chi={args.chi} step (np.dot per pair, scipy.linalg.svd per block, host '
                                         f'truncation) at the best BLAS thread count of the sweep, median of {cb["reps_best"]} '
                                         f'after one warm call',
                               **{k: cb[k] for k in ('sweep_seconds_per_step', 'best_threads', 'all_threads', 'seconds_all',
                                                     'split_best', 'single_thread', 'blas', 'host_seconds_used')}}
        out['speedup_vs_cpu'] = round(value / out['cpu_baseline']['value'], 2)
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            threads = out.get('cpu_baseline', {}).get('cores', 16)
            out['roofline_svd_fullrank'] = fullrank_svd_roofline(bb, res['shapes'], threads)
        except Exception as e:
            out['roofline_svd_fullrank'] = {'error': repr(e)}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
