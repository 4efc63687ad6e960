"""The N > 1 path of bench.py's step on ONE GPU: rank r of a world of 2 / 4 computes its own units into the rank-major
pool; the collective is replaced by a stand-in that fills the other ranks' segments from a single-rank reference run
(exactly what all_gather_into_tensor delivers).  Checks the GPU-side offsets / views of the sharded layout, that every
rank ends with the same theta, singular values and truncation as the unsharded step, and that the ranks' units
partition the work.  (The collective itself is covered by the gloo test on CPU and BENCH_FORCE_DIST on the GPU box.)"""
import numpy as np
import pytest
import torch

import bench
from cyten_amd import sharding, workloads as wl

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_step_matches_single_rank(bb, monkeypatch, world):
    A, B = wl.config_u1_mps(256)
    chi_max = 200
    ref_step = bench.ThetaStep(bb, A, B, chi_max)
    ref = ref_step.step(timed_gemm=False)
    ref_theta = [bb.to_numpy(x).reshape(-1) for x in ref['theta'].blocks]
    ref_usv = [np.concatenate([bb.to_numpy(u).reshape(-1), bb.to_numpy(s), bb.to_numpy(v).reshape(-1)]) for u, s, v in ref['usv']]
    ref_S = np.concatenate([bb.to_numpy(s) for _, s, _ in ref['usv']])
    calls = []

    def fake_gather(pool, layout, rank, group=None):
        # phase 1 pools hold theta blocks, phase 2 pools hold U | S | Vh per sector
        data = ref_theta if len(layout.sizes) == len(ref_theta) and len(calls) % 2 == 0 else ref_usv
        assert layout.world == world and len(layout.sizes) == len(data)
        mine = set(layout.local_units(rank))
        host = pool.cpu().numpy()
        for u, d in enumerate(data):
            seg = host[layout.offset[u]:layout.offset[u] + len(d)]
            if u in mine:  # what this rank computed itself must already be the reference (up to SVD sign freedom)
                if data is ref_theta:
                    np.testing.assert_allclose(seg, d, rtol=0, atol=1e-10 * max(1.0, np.abs(d).max()))
            else:
                seg[:] = d
        pool.copy_(torch.from_numpy(host).to(pool.device))
        calls.append((rank, len(mine)))
        return pool

    monkeypatch.setattr(sharding, 'allgather_pool', fake_gather)
    owned_gemm, owned_svd = [], []
    for rank in range(world):
        calls.clear()
        st = bench.ThetaStep(bb, A, B, chi_max, rank, world)
        res = st.step(timed_gemm=False)
        assert len(calls) == 2
        owned_gemm.append(calls[0][1])
        owned_svd.append(calls[1][1])
        for x, want in zip(res['theta'].blocks, ref_theta):
            np.testing.assert_allclose(bb.to_numpy(x).reshape(-1), want, rtol=0, atol=1e-10 * max(1.0, np.abs(want).max()))
        S = np.concatenate([bb.to_numpy(s) for _, s, _ in res['usv']])
        assert np.abs(S - ref_S).max() <= 1e-10 * ref_S.max()
        assert abs(res['err'] - ref['err']) <= 1e-10 * (ref['err'] + ref['new_norm'])
        assert abs(res['new_norm'] - ref['new_norm']) <= 1e-10 * ref['new_norm']
        assert sum(m.n for m in res['masks']) == sum(m.n for m in ref['masks']) == chi_max
        assert res['imbalance_gemm'] >= 1.0 and res['imbalance_svd'] >= 1.0
    assert sum(owned_gemm) == len(ref_theta) and sum(owned_svd) == len(ref_usv) and min(owned_gemm) > 0 and min(owned_svd) > 0
