import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cyten_amd.block_backend import HipBlockBackend
bb = HipBlockBackend('cuda:0')
rng = np.random.default_rng(12345)
shapes = [(1, 1), (2, 1), (1, 2), (3, 3), (7, 5), (5, 7), (64, 64), (63, 64), (64, 63), (128, 64), (64, 128), (128, 1),
          (1, 128), (127, 33), (33, 127), (17, 17), (40, 40), (128, 63)]
mats = [rng.standard_normal(s) for s in shapes]
low = rng.standard_normal((50, 4)) @ rng.standard_normal((4, 33))
dup = rng.standard_normal((30, 12)); dup[:, 5] = dup[:, 2]; dup[:, 9] = 0.0
zero_rows = rng.standard_normal((40, 20)); zero_rows[10:30] = 0.0
mats += [low, low.T.copy(), dup, dup.T.copy(), zero_rows, np.zeros((9, 5)), np.zeros((5, 9)), np.eye(33), np.ones((20, 31)),
         np.diag(np.r_[np.ones(10), np.zeros(7)]), 1e-200 * rng.standard_normal((12, 12)), 1e200 * rng.standard_normal((12, 30))]
res, info = bb.matrix_svd_batched([bb.as_block(m) for m in mats], return_info=True)
for i, (m, (u, s, vh)) in enumerate(zip(mats, res)):
    u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
    nrm = max(np.abs(m).max(), 1e-300)
    rec = np.abs((u * s) @ vh - m).max() / nrm
    ou = np.abs(u.T @ u - np.eye(u.shape[1])).max()
    ov = np.abs(vh @ vh.T - np.eye(vh.shape[0])).max()
    sref = np.linalg.svd(m, compute_uv=False)
    ds = np.abs(s - sref).max() / max(sref.max(), 1e-300)
    flag = '' if max(rec, ou, ov, ds) < 1e-10 else '  <<<<'
    print(i, m.shape, 'recon %.1e orthoU %.1e orthoV %.1e dS %.1e' % (rec, ou, ov, ds), flag)
print('--- 400 batch')
many = [rng.standard_normal((int(rng.integers(1, 41)), int(rng.integers(1, 41)))) for _ in range(400)]
res = bb.matrix_svd_batched([bb.as_block(m) for m in many])
bad = 0
for i, (m, (u, s, vh)) in enumerate(zip(many, res)):
    u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
    rec = np.abs((u * s) @ vh - m).max() / np.abs(m).max()
    ou = np.abs(u.T @ u - np.eye(u.shape[1])).max(); ov = np.abs(vh @ vh.T - np.eye(vh.shape[0])).max()
    ds = np.abs(s - np.linalg.svd(m, compute_uv=False)).max()
    if max(rec, ou, ov, ds) > 1e-10:
        bad += 1
        if bad < 8: print(i, m.shape, 'recon %.1e orthoU %.1e orthoV %.1e dS %.1e' % (rec, ou, ov, ds))
print('bad', bad)
print('--- views')
big = rng.standard_normal((90, 150)); B = bb.as_block(big)
views = [bb.get_item(B, (slice(3, 60), slice(7, 47))), bb.permute_axes(bb.get_item(B, (slice(0, 30), slice(1, 100))), [1, 0]), B]
refs = [big[3:60, 7:47], big[0:30, 1:100].T, big]
for m, (u, s, vh) in zip(refs, bb.matrix_svd_batched(views)):
    u, s, vh = bb.to_numpy(u), bb.to_numpy(s), bb.to_numpy(vh)
    print(m.shape, 'recon %.1e' % (np.abs((u * s) @ vh - m).max()), 'dS %.1e' % np.abs(s - np.linalg.svd(m, compute_uv=False)).max())
