"""Host profile (cProfile) of the bench step -- compose plan, sector plan, grouped GEMM, combine_legs, batched SVD, device
truncation, kept-column gather -- for the U(1) or the U(1)xU(1) chi=4096 theta:  python3 scripts/step_profile.py [u1|u1u1] [chi]"""
import cProfile, io, pstats, sys, time
sys.path.insert(0, '.')
import torch
import bench
from cyten_amd import workloads as wl
from cyten_amd.block_backend import HipBlockBackend

sym = sys.argv[1] if len(sys.argv) > 1 else 'u1u1'
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
bb = HipBlockBackend('cuda:0')
A, B = (wl.config_u1u1_mps(chi) if sym == 'u1u1' else wl.config_u1_mps(chi))
st = bench.ThetaStep(bb, A, B, chi)
for _ in range(2):
    st.step(timed=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    st.step(timed=False)
torch.cuda.synchronize()
pr.disable()
print(f'[step-profile] {sym} chi={chi}: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per step under cProfile')
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(40)
print(out.getvalue()[:8000])
