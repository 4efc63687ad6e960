"""cProfile of the host side of bench.ThetaStep.step (development aid)."""
import sys, cProfile, pstats
sys.path.insert(0, '.')
import bench
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import workloads as wl
bb = HipBlockBackend('cuda:0')
A, B = wl.config_u1_mps(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
st = bench.ThetaStep(bb, A, B, 4096)
for _ in range(2):
    st.step(); bb.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    st.step(); bb.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
