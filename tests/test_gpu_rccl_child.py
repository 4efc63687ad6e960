"""The RCCL code path of the sharded step on the GPU box: a FRESH child process (the launcher rule: the process group is
created before anything else touches the GPU) initialises `nccl` (= RCCL on ROCm) with world size 1 and runs bench.py's
sharded theta step -- process-group init on the device, the in-place `all_gather_into_tensor` of the singular values and of
the kept factors, barrier, all_reduce of the step time, the per-rank report -- and must reproduce the plain single-process
result.  (World size 2 is covered over gloo on the CPU, tests/test_distributed.py; N = 2, 4, 8 on hardware is the driver's run.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--chi', '256', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extras']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [x for x in r.stdout.strip().splitlines() if x.startswith('{')][-1]
    return json.loads(line)


def test_sharded_step_over_rccl_with_one_rank_matches_the_plain_run():
    plain = _run({})
    rccl = _run({'BENCH_FORCE_DIST': '1', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29577', 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'})
    assert plain['rccl_ranks'] == 0 and rccl['rccl_ranks'] == 1 and rccl['n_gpus'] == 1
    for key in ('err', 'new_norm', 'kept'):
        a, b = plain['truncation'][key], rccl['truncation'][key]
        assert abs(a - b) <= 1e-10 * max(1.0, abs(a)), (key, a, b)
    assert rccl['config']['svd_blocks'] == plain['config']['svd_blocks']
    rep = rccl['ranks']
    assert len(rep) == 1 and rep[0]['rank'] == 0 and rep[0]['sectors'] == rccl['config']['svd_blocks']
    assert rep[0]['svd_ms'] > 0 and rep[0]['gemm_ms'] > 0 and rep[0]['collectives_ms'] > 0      # the two all_gathers really ran
