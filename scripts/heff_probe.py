"""Run a few H_eff matvecs (for rocprofv3 --kernel-trace --stats)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cyten_amd.block_backend import HipBlockBackend
from cyten_amd import krylov, workloads as wl
from helpers import to_device_tensor
bb = HipBlockBackend('cuda:0')
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = wl.config_heff(chi, 5, seed=11)
dev = {k: to_device_tensor(bb, v) for k, v in cfg.items()}
H = krylov.HEffective(bb, dev['LP'], dev['W1'], dev['W2'], dev['RP'])
for _ in range(4):
    H.matvec(dev['theta']); bb.synchronize()
print('probe done')
