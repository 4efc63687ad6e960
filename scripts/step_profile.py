"""cProfile of the host side of one bench step (development aid)."""
import sys, cProfile, pstats, time
sys.path.insert(0, '.')
import bench
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import workloads as wl
bb = HipBlockBackend('cuda:0')
A, B = (wl.config_u1u1_mps(4096) if len(sys.argv) > 1 and sys.argv[1] == "u1u1" else wl.config_u1_mps(4096))
st = bench.ThetaStep(bb, A, B, 4096)
for _ in range(3):
    st.step(); bb.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    st.step()
bb.synchronize()
print('step %.2f ms' % (200 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    st.step()
bb.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(24)
