"""Grouped-GEMM timing of the BASELINE block lists (HIP events on the launch stream), for A/B runs and the PMC passes:
  python scripts/gemm_lists.py [u1] [u1u1] [uniform] [u1_2048] [reps=N]
prints ms per launch, TFLOP/s and the fraction of the 78.6 TFLOP/s f64 MFMA peak."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import bench
from cyten_amd import abelian as ab
from cyten_amd import workloads as wl
from cyten_amd.block_backend import HipBlockBackend

bb = HipBlockBackend('cuda:0')
args = [a for a in sys.argv[1:] if '=' not in a] or ['u1', 'u1u1', 'uniform']
reps = int(dict(a.split('=') for a in sys.argv[1:] if '=' in a).get('reps', 20))


def time_plan(name, gemm, check=None):
    for _ in range(3):
        gemm.run()
    t = bench.Timer(bb.ctx, reps)
    for _ in range(reps):
        with t:
            gemm.run()
    bb.synchronize()
    ms = np.array(t.ms())
    tf = gemm.flops / (ms.mean() * 1e-3) / 1e12
    print(f'[gemm] {name}: {ms.mean() * 1e3:.1f} us (min {ms.min() * 1e3:.1f}) -> {tf:.2f} TFLOP/s = {tf / 78.6:.3f} of peak; '
          f'{gemm.flops / 1e9:.2f} GFLOP, {gemm.bytes / 1e6:.0f} MB algorithmic', flush=True)


def theta_list(name, A, B):
    a, b = ab.AbelianTensor.from_spec(bb, A), ab.AbelianTensor.from_spec(bb, B)
    plan = ab.compose_plan(a, b, 1)
    a2, b2 = ab._compose_operands(bb, a, b, 1, plan)
    groups = [[(a2[i], b2[j]) for i, j in g] for g in plan.pairs]
    gemm = bb.make_gemm_plan(groups)
    time_plan(f'{name}: {len(groups)} problems / {sum(len(g) for g in groups)} GEMMs', gemm)
    # spot check of the largest result block against numpy
    k = int(np.argmax([np.prod(s) for s in plan.res_shapes]))
    want = sum(bb.to_numpy(x) @ bb.to_numpy(y) for x, y in groups[k])
    got = bb.to_numpy(gemm.outs[k])
    print(f'        largest block {want.shape}: max err {np.abs(got - want).max() / np.abs(want).max():.1e}')
    gemm.destroy()


if 'u1' in args:
    theta_list('U(1) chi=4096 theta', *wl.config_u1_mps(4096))
if 'u1u1' in args:
    theta_list('U(1)xU(1) chi=4096 theta', *wl.config_u1u1_mps(4096))
if 'u1_2048' in args:
    theta_list('U(1) chi=2048 theta', *wl.config_u1_mps(2048))
if 'uniform' in args:
    rng = np.random.default_rng(0)
    x, y = bb.as_block(rng.standard_normal((4096, 4096))), bb.as_block(rng.standard_normal((4096, 4096)))
    time_plan('uniform 4096^3', bb.make_gemm_plan([[(x, y)]]))
