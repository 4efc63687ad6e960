"""The N > 1 path of bench.py's step on ONE GPU: rank r of a world of 2 / 4 contracts, combines and decomposes only the
theta blocks of its own coupled-charge sectors; the two collectives (singular values, kept factors) are replaced by a
stand-in that fills the other ranks' segments from a single-rank reference run (exactly what all_gather_into_tensor
delivers).  Checks the GPU-side offsets / views of the sharded layouts, that every rank ends with the same singular
values, truncation and truncated factors as the unsharded step, that a rank never touches foreign sectors, and that
the ranks' sectors partition the work.  (The collectives themselves are covered by the gloo test on CPU and
BENCH_FORCE_DIST on the GPU box.)"""
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
    ref = bench.ThetaStep(bb, A, B, chi_max).step(timed=False)
    n_sec = len(ref['shapes'])
    ref_S = [bb.to_numpy(s) for _, s, _ in ref['usv']]
    ref_kept = [np.concatenate([bb.to_numpy(u).reshape(-1), bb.to_numpy(s), bb.to_numpy(v).reshape(-1)]) for u, s, v in ref['kept']]
    ref_prod = [(bb.to_numpy(u) * bb.to_numpy(s)) @ bb.to_numpy(v) for u, s, v in ref['kept']]
    calls = []

    def fake_gather(pool, layout, rank, group=None):
        data = ref_S if len(calls) == 0 else ref_kept      # first collective: singular values; second: kept U | S | Vh
        assert layout.world == world and len(layout.sizes) == len(data) == n_sec
        mine = set(layout.local_units(rank))
        host = pool.cpu().numpy()
        for u, d in enumerate(data):
            seg = host[layout.offset[u]:layout.offset[u] + len(d)]
            if u in mine:
                if data is ref_S:   # what this rank computed itself must already be the reference
                    np.testing.assert_allclose(seg, d, rtol=0, atol=1e-10 * max(1.0, np.abs(d).max()))
            else:
                seg[:] = d
        pool.copy_(torch.from_numpy(host).to(pool.device))
        calls.append((rank, len(mine)))
        return pool

    monkeypatch.setattr(sharding, 'allgather_pool', fake_gather)
    owned = []
    for rank in range(world):
        calls.clear()
        res = bench.ThetaStep(bb, A, B, chi_max, rank, world).step(timed=False)
        assert len(calls) == 2
        owned.append(calls[0][1])
        assert len(res['local_sectors']) == calls[0][1] == len(res['mv'].blocks)
        # theta stays local: this rank only holds the theta blocks of its own sectors
        assert len(res['theta'].blocks) < len(ref['theta'].blocks)
        assert abs(res['err'] - ref['err']) <= 1e-10 * (ref['err'] + ref['new_norm'])
        assert abs(res['new_norm'] - ref['new_norm']) <= 1e-10 * ref['new_norm']
        assert int(res['kept_n'].sum()) == int(ref['kept_n'].sum()) == chi_max
        for u, (uu, ss, vv) in enumerate(res['kept']):     # truncated factors of EVERY sector are addressable here
            got = (bb.to_numpy(uu) * bb.to_numpy(ss)) @ bb.to_numpy(vv)
            np.testing.assert_allclose(got, ref_prod[u], rtol=0, atol=1e-10 * max(1.0, np.abs(ref_prod[u]).max()))
        assert res['imbalance_svd'] >= 1.0
    assert sum(owned) == n_sec and min(owned) > 0
