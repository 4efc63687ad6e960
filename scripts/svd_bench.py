"""Timing probe for the batched SVD (warm, repeated): prints ms per call and sweeps."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import workloads as wl

bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(0)


def run(name, mats, reps=3, check=True):
    blocks = [bb.as_block(m) for m in mats]
    res, info = bb.matrix_svd_batched(blocks, return_info=True)   # warm (allocations)
    bb.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        res = bb.matrix_svd_batched(blocks)
        bb.synchronize()
        ts.append(time.perf_counter() - t0)
    worst = 0.0
    if check:
        for m, (U, S, Vh) in zip(mats, res):
            U, S, Vh = bb.to_numpy(U), bb.to_numpy(S), bb.to_numpy(Vh)
            nrm = np.linalg.norm(m)
            sref = np.linalg.svd(m, compute_uv=False)
            k = len(S)
            worst = max(worst, np.abs(S - sref).max() / nrm, np.abs((U * S) @ Vh - m).max() / nrm,
                        np.abs(U.T @ U - np.eye(k)).max(), np.abs(Vh @ Vh.T - np.eye(k)).max())
    flops = wl.svd_nominal_flops([m.shape for m in mats])
    print(f'[svdbench] {name}: {len(mats)} blocks, best {1e3 * min(ts):.2f} ms (median {1e3 * np.median(ts):.2f}), '
          f'{flops / min(ts) / 1e9:.0f} nominal GFLOP/s, sweeps={info}, worst err {worst:.1e}', flush=True)


what = sys.argv[1:] or ['cfg2', 'full3', 'theta4096']
if 'cfg2' in what:
    sizes = [8, 29, 74, 150, 246, 335, 360, 335, 246, 150, 74, 29, 8]
    run('cfg2 full-rank', [rng.standard_normal((s, s)) for s in sizes])
if 'full3' in what:
    run('full-rank 1442/1236/721', [rng.standard_normal((s, s)) for s in (1442, 1236, 721)])
if 'single' in what:  # what the rank owning the largest block of a sharded step works on
    a = rng.standard_normal((1442, 721)) @ rng.standard_normal((721, 1442))
    run('single rank-deficient 1442', [a], reps=2)
    run('single full-rank 1442', [rng.standard_normal((1442, 1442))], reps=2)
if 'tall' in what:
    run('tall 9216x256 x4', [rng.standard_normal((9216, 256)) for _ in range(4)])
if 'theta4096' in what:
    # the SVD block list of the chi=4096 U(1) theta: every block has rank ~ half its size
    leg = wl.u1_leg(4096, 2.0)
    mult = {int(q[0]): int(m) for q, m in zip(leg.sectors, leg.mults)}
    mats = []
    for q in sorted(mult):
        rows = mult.get(q - 1, 0) + mult.get(q + 1, 0)
        if rows == 0:
            continue
        a = rng.standard_normal((rows, mult[q]))
        b = rng.standard_normal((mult[q], rows))
        mats.append(a @ b)
    run('theta chi=4096 (rank-deficient)', mats, reps=2)
