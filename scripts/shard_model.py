"""Strong-scaling table of the sector-sharded step, measured by emulation on ONE device: for N in 1, 2, 4, 8 the sectors of
the chi=4096 U(1) theta are LPT-assigned to N ranks exactly as bench.py does it, every rank's share of the step (its
GEMMs, combine, batched SVD, kept-column gather) is timed alone on the device, and the step time of the N-GPU run is
modelled as max over ranks + the two collectives priced at xGMI figures (S: a few KB, latency only; kept factors:
bytes / 100 GB/s).  What it cannot see: RCCL launch latencies beyond that and 8 processes sharing a host.  DESIGN.md 5."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from cyten_amd import sharding, workloads as wl
from cyten_amd.block_backend import HipBlockBackend

bb = HipBlockBackend('cuda:0')
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A, B = wl.config_u1_mps(chi)
sharding.allgather_pool = lambda pool, layout, rank, group=None: pool      # the collectives are priced, not run
rows = []
for world in (1, 2, 4, 8):
    times, kept_bytes = [], 0
    for rank in range(world):
        st = bench.ThetaStep(bb, A, B, chi, rank, world)
        for _ in range(2):
            st.step(timed=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            res = st.step(timed=False)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 3 * 1e3)
        kept_bytes = sum(int(np.prod(x.shape)) * 8 for trip in res['kept'] for x in trip)
    coll_ms = 0.0 if world == 1 else 2 * 0.03 + kept_bytes * (world - 1) / world / 100e9 * 1e3
    rows.append((world, max(times), coll_ms, times))
    print(f'[shard] N={world}: slowest rank {max(times):.1f} ms (ranks: {", ".join(f"{t:.1f}" for t in times)}), collectives ~{coll_ms:.2f} ms '
          f'-> step ~{max(times) + coll_ms:.1f} ms, speed-up {rows[0][1] / (max(times) + coll_ms):.2f}x', flush=True)
