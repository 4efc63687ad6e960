"""A short leg of each randomised soak of scripts/ (svd_fuzz / ops_fuzz / tensor_fuzz / trunc_fuzz / callers_fuzz) as a regression test: fixed seeds, a few
rounds each, one child process at a time.  The long runs are made by hand; what they found is pinned by the
dedicated tests of test_gpu_decomp.py / test_gpu_complex.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('script,count,seed', [
    ('svd_fuzz.py', 6, 5),
    ('ops_fuzz.py', 150, 5),
    ('tensor_fuzz.py', 60, 5),
    ('trunc_fuzz.py', 1500, 5),
    ('callers_fuzz.py', 8, 5),
])
def test_soak_leg(script, count, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', script), str(count), str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    tail = '\n'.join(r.stdout.strip().splitlines()[-8:])
    assert r.returncode == 0, f'{script}: rc {r.returncode}\n{tail}\n{r.stderr[-2000:]}'
    assert ('done:' in tail or 'rounds,' in tail) and ' 0 failures' in tail, tail
