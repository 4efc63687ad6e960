"""World-size-2 gloo test (CPU) of the sector-sharded theta step (bench.py's N > 1 path): coupled-charge sectors are
LPT-assigned to the ranks, a rank contracts / combines / decomposes ONLY the theta blocks of its own sectors, one small
all_gather makes the singular values global, the truncation is the same selection on every rank, and one all_gather
of the KEPT U / S / Vh leaves the truncated factors of every sector on every rank.  The per-unit compute is the
oracle's numpy ops (this tests the N>1 plumbing and ownership logic, not the kernels).  Also: the launcher decision of
`python bench.py --gpus N` (no GPU needed: --dry-run) and the round-ownership schedule of the intra-block split."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cyten_amd import abelian as ab, sharding, workloads as wl
        from numpy_backend import NumpyGroupedBackend
        from oracle import abelian_ref as ref
        nb = NumpyGroupedBackend()
        chi_max = 60
        A, B = wl.config_u1_mps(96)                      # same seed on every rank: replicated operands
        a, b = ab.AbelianTensor.from_spec(nb, A), ab.AbelianTensor.from_spec(nb, B)
        plan = ab.compose_plan(a, b, 1)
        sp = sharding.theta_sector_plan(plan, a, 2, world)
        mine_sec = sp.layout.local_units(rank)
        mine_blk = sp.blocks_of(mine_sec)
        a2, b2 = ab._compose_operands(nb, a, b, 1, plan)
        blocks = [sum(a2[i] @ b2[j] for i, j in plan.pairs[u]).reshape(plan.res_shapes[u]) for u in mine_blk]
        theta = ab.AbelianTensor(a.symmetry, plan.legs, blocks, plan.res_block_inds[mine_blk], 2)
        mv = ab.combine_legs_to_matrix(nb, theta, 2)
        ok = [m.shape for m in mv.blocks] == [sp.shapes[u] for u in mine_sec]
        usv = [np.linalg.svd(m, full_matrices=False) for m in mv.blocks]
        # S pool: rank-major, one all_gather
        s_pool = torch.zeros(sp.s_layout.total, dtype=torch.float64)
        for (_, s, _), u in zip(usv, mine_sec):
            s_pool[sp.s_layout.offset[u]:sp.s_layout.offset[u] + len(s)] = torch.from_numpy(s)
        sharding.allgather_pool(s_pool, sp.s_layout, rank)
        ks = [min(s) for s in sp.shapes]
        S = [s_pool[sp.s_layout.offset[u]:sp.s_layout.offset[u] + ks[u]].numpy() for u in range(len(ks))]
        mask, err, new_norm = ab.truncation_selection(np.concatenate(S), chi_max=chi_max)
        offs = np.concatenate([[0], np.cumsum(ks)])
        masks = [mask[offs[u]:offs[u + 1]] for u in range(len(ks))]
        kept_n = np.array([int(m.sum()) for m in masks])
        ksz = np.array([sp.shapes[u][0] * kept_n[u] + kept_n[u] + kept_n[u] * sp.shapes[u][1] for u in range(len(ks))])
        k_lay = sharding.layout_for_owner(ksz, sp.layout.owner, world)
        k_pool = torch.zeros(k_lay.total, dtype=torch.float64)
        for (U, s, Vh), u in zip(usv, mine_sec):
            m = masks[u]
            flat = np.concatenate([U[:, m].reshape(-1), s[m], Vh[m, :].reshape(-1)])
            k_pool[k_lay.offset[u]:k_lay.offset[u] + len(flat)] = torch.from_numpy(flat)
        sharding.allgather_pool(k_pool, k_lay, rank)
        # every rank now reconstructs the truncated theta of EVERY sector and compares with the unsharded oracle
        oracle = ref.theta_tdot_svd(A, B, chi_max=chi_max)
        ok = ok and abs(err - oracle['err']) <= 1e-12 and abs(new_norm - oracle['new_norm']) <= 1e-10 * oracle['new_norm']
        ok = ok and np.abs(np.concatenate(S) - oracle['S_all']).max() < 1e-10 * oracle['S_all'].max()
        ooffs = np.concatenate([[0], np.cumsum([len(s) for _, s, _ in oracle['usv']])])
        for u in range(len(ks)):
            mrow, ncol = sp.shapes[u]
            c, o = int(kept_n[u]), int(k_lay.offset[u])
            flat = k_pool[o:o + ksz[u]].numpy()
            U, s, Vh = flat[:mrow * c].reshape(mrow, c), flat[mrow * c:mrow * c + c], flat[mrow * c + c:].reshape(c, ncol)
            Uo, so, Vo = oracle['usv'][u]
            mo = oracle['mask'][ooffs[u]:ooffs[u + 1]]
            want = (Uo[:, mo] * so[mo]) @ Vo[mo, :]
            ok = ok and np.abs((U * s) @ Vh - want).max() <= 1e-10 * max(1.0, np.abs(oracle['matrices'][u]).max())
        ok = ok and len(set(sp.layout.owner.tolist())) == world and len(mine_sec) > 0
        ok = ok and sorted(np.concatenate([sp.blocks_of(sp.layout.local_units(r)) for r in range(world)]).tolist()) == list(range(len(plan.pairs)))
        # every rank must hold bit-identical pools
        gathered = [torch.zeros_like(k_pool) for _ in range(world)]
        dist.all_gather(gathered, k_pool)
        ok = ok and all(torch.equal(gathered[0], g) for g in gathered)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_sector_sharding_gloo_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_allgather_pool_world1_is_identity():
    sys.path.insert(0, ROOT)
    from cyten_amd import sharding
    lay = sharding.make_layout([5, 7], [1.0, 2.0], 1)
    pool = torch.arange(lay.total, dtype=torch.float64)
    assert sharding.allgather_pool(pool, lay, 0) is pool


def _dry(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args, '--dry-run'], env=env, capture_output=True,
                         text=True, check=True)
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_launcher_starts_n_ranks_itself():
    """`python bench.py --gpus N` with no torch.distributed environment must start N ranks."""
    d = _dry(['--gpus', '8', '--steps', '3', '--warmup', '1'])
    assert d['role'] == 'launcher' and d['n_gpus'] == 8
    cmd = d['launch']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=8' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    tail = cmd[cmd.index(os.path.join(ROOT, 'bench.py')) + 1:]
    assert tail == ['--gpus', '8', '--steps', '3', '--warmup', '1']      # the ranks get the same arguments
    # one GPU: no launcher; under torch.distributed.run (RANK / WORLD_SIZE set): this process is a rank
    assert _dry(['--gpus', '1'])['launch'] is None
    d = _dry(['--gpus', '4'], {'RANK': '2', 'WORLD_SIZE': '4', 'LOCAL_RANK': '2'})
    assert d['launch'] is None and d['role'] == 'rank'


def _jacobi_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import column_split_spec as spec
        rng = np.random.default_rng(5)                      # same data on every rank
        nv, length = 64, 192                                # 4 row blocks of 16, three 64-column chunks
        A = rng.standard_normal((nv, 40)) @ rng.standard_normal((40, length)) + 1e-3 * rng.standard_normal((nv, length))
        w_slabs = spec.column_slabs(length, world)
        j_slabs = spec.column_slabs(nv, world)
        (w0, w1), (j0, j1) = w_slabs[rank], j_slabs[rank]
        W, J, sweeps = spec.distributed_block_jacobi(A[:, w0:w1], np.eye(nv)[:, j0:j1])
        # gather the slabs (test only) and check: rows orthogonal, norms = singular values, J orthogonal, J A = W
        parts = [None] * world
        dist.all_gather_object(parts, (W, J))
        Wf = np.concatenate([p[0] for p in parts], axis=1)
        Jf = np.concatenate([p[1] for p in parts], axis=1)
        s = np.sort(np.linalg.norm(Wf, axis=1))[::-1]
        sref = np.linalg.svd(A, compute_uv=False)
        G = Wf @ Wf.T
        offd = np.abs(G - np.diag(np.diag(G))).max() / np.abs(G).max()
        ok = sweeps > 0 and np.abs(s - sref).max() <= 1e-10 * sref[0] and offd <= 1e-12
        ok = ok and np.abs(Jf @ Jf.T - np.eye(nv)).max() <= 1e-12 and np.abs(Jf @ A - Wf).max() <= 1e-10 * np.abs(A).max()
        # ownership: the slabs partition the columns in whole 64-column chunks; the schedule meets every pair once
        ok = ok and w_slabs[0][0] == 0 and w_slabs[-1][1] == length and all(a[1] == b[0] for a, b in zip(w_slabs, w_slabs[1:]))
        ok = ok and all((b - a) % spec.CHUNK == 0 for a, b in w_slabs)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_column_split_block_jacobi_gloo_world2():
    """the intra-block split (columns of ONE block's rows over the ranks, one all_reduce of the pair
    Gram matrices per round, no row exchange) -- schedule / ownership logic and the algorithm itself, world size 2."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_jacobi_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_round_robin_schedule_and_slabs():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import column_split_spec as spec
    for nb in (2, 4, 12, 48):
        rounds = spec.round_robin_schedule(nb)
        assert len(rounds) == nb - 1 and all(len(r) == nb // 2 for r in rounds)
        seen = [p for r in rounds for p in r]
        assert len(set(seen)) == nb * (nb - 1) // 2 and all(p < q for p, q in seen)        # every pair exactly once
        assert all(len({b for pq in r for b in pq}) == nb for r in rounds)                    # disjoint within a round
    assert spec.column_slabs(1472, 3) == [(0, 448), (448, 960), (960, 1472)]
    assert spec.split_round_model(1) == 40.0 and spec.split_round_model(8) > 36.0    # the split does not pay
