"""World-size-2 gloo test (CPU) of the sector-sharded path: LPT assignment, pool layout, ONE
all_gather per phase, every rank ends with the complete and identical block list.  The per-unit
compute is the oracle's numpy ops (this tests the N>1 plumbing, not the kernels)."""
import os
import socket
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
        A, B = wl.config_u1_mps(96)                      # same seed on every rank: replicated operands
        a, b = ab.AbelianTensor.from_spec(nb, A), ab.AbelianTensor.from_spec(nb, B)
        plan = ab.compose_plan(a, b, 1)
        sizes = [int(np.prod(s)) for s in plan.res_shapes]
        costs = [sum(2.0 * np.prod(s) * a.blocks[i].shape[-1] for i, _ in g) for g, s in zip(plan.pairs, plan.res_shapes)]
        lay = sharding.make_layout(sizes, costs, world)
        pool = torch.zeros(lay.total, dtype=torch.float64)
        a2, b2 = ab._compose_operands(nb, a, b, 1, plan)
        for u in lay.local_units(rank):                  # this rank's GEMM problems only
            acc = sum(a2[i] @ b2[j] for i, j in plan.pairs[u])
            pool[lay.offset[u]:lay.offset[u] + sizes[u]] = torch.from_numpy(acc.reshape(-1))
        sharding.allgather_pool(pool, lay, rank)
        blocks = [pool[lay.offset[u]:lay.offset[u] + sizes[u]].numpy().reshape(plan.res_shapes[u]) for u in range(len(sizes))]
        want, bi, _ = ref.compose(A, B, 1)
        ok = np.array_equal(plan.res_block_inds, bi) and all(np.abs(x - y).max() < 1e-12 for x, y in zip(blocks, want))
        # second phase: sector blocks of the SVD, sharded by nominal flops
        theta = ab.AbelianTensor(a.symmetry, plan.legs, blocks, plan.res_block_inds, 2)
        mv = ab.combine_legs_to_matrix(nb, theta, 2)
        shapes = [m.shape for m in mv.blocks]
        lay2 = sharding.make_layout([min(s) for s in shapes], [4.0 * max(s) * min(s) ** 2 + 8.0 * min(s) ** 3 for s in shapes], world)
        pool2 = torch.zeros(lay2.total, dtype=torch.float64)
        for u in lay2.local_units(rank):
            s = np.linalg.svd(mv.blocks[u], compute_uv=False)
            pool2[lay2.offset[u]:lay2.offset[u] + len(s)] = torch.from_numpy(s)
        sharding.allgather_pool(pool2, lay2, rank)
        S_all = np.concatenate([pool2[lay2.offset[u]:lay2.offset[u] + min(shapes[u])].numpy() for u in range(len(shapes))])
        oracle = ref.theta_tdot_svd(A, B)
        ok = ok and np.abs(S_all - oracle['S_all']).max() < 1e-10 * oracle['S_all'].max()
        ok = ok and len(set(lay.owner.tolist())) == world and len(lay.local_units(rank)) > 0
        # every rank must hold bit-identical pools
        gathered = [torch.zeros_like(pool2) for _ in range(world)]
        dist.all_gather(gathered, pool2)
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
