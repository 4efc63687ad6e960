"""cProfile of the host side of bench.py's step (which Python functions the milliseconds between the kernels go to).
    python scripts/host_profile.py [u1|u1u1] [steps]"""
import cProfile
import pstats
import sys

sys.path.insert(0, '.')
import torch

import bench
from cyten_amd import workloads as wl
from cyten_amd.block_backend import HipBlockBackend

sym = sys.argv[1] if len(sys.argv) > 1 else 'u1u1'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bb = HipBlockBackend('cuda:0')
A, B = wl.config_u1_mps(4096) if sym == 'u1' else wl.config_u1u1_mps(4096)
st = bench.ThetaStep(bb, A, B, 4096)
for _ in range(3):
    st.step(timed=False)
torch.cuda.synchronize()
import gc
gc.collect()
gc.freeze()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    st.step(timed=False)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr)
ps.sort_stats('cumulative').print_stats(45)
