"""First-light check of the C-ABI kernels on a real MI355X (run through gpurun).

Not part of the test-suite: a quick numerics + timing probe used while bringing kernels up.
"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from cyten_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
ctx = C.c_void_p()
L.check(lib.cyb_ctx_create(C.byref(ctx), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def dptr(t):
    return C.c_void_p(t.data_ptr())


def peak():
    for nacc in (1, 2, 4, 8):
        for wps in (1, 2, 4):
            tf = C.c_double()
            ms = C.c_double()
            L.check(lib.cyb_mfma_f64_peak(ctx, 400000, nacc * 100 + wps, C.byref(tf), C.byref(ms)))
            cyc = 256 * 4 * wps * 400000 / (ms.value * 1e-3) 
            print(f'[peak] v_mfma_f64_16x16x4_f64 {nacc} acc, {wps} wave/SIMD: {tf.value:.2f} TFLOP/s ({ms.value:.3f} ms; '
                  f'{2.4e9 / (400000 * wps / (ms.value * 1e-3)):.1f} cyc/MFMA/SIMD at 2.4 GHz)')


def gemm_case(M, N, Ks, ta=False, tb=False, seed=0):
    rng = np.random.default_rng(seed)
    probs = (L.GemmProb * 1)()
    segs = (L.GemmSeg * len(Ks))()
    keep = []
    ref = np.zeros((M, N))
    for s, K in enumerate(Ks):
        a = rng.standard_normal((M, K))
        b = rng.standard_normal((K, N))
        ref += a @ b
        ta_ = ta if s % 2 == 0 else not ta  # mix layouts across segments
        A = torch.from_numpy(np.ascontiguousarray(a.T if ta_ else a)).to(dev)
        B = torch.from_numpy(np.ascontiguousarray(b.T if tb else b)).to(dev)
        keep += [A, B]
        segs[s].A = A.data_ptr()
        segs[s].B = B.data_ptr()
        segs[s].K = K
        segs[s].a_rs, segs[s].a_cs = (1, M) if ta_ else (K, 1)
        segs[s].b_rs, segs[s].b_cs = (1, K) if tb else (N, 1)
    Cm = torch.full((M, N + 3), 7.0, dtype=torch.float64, device=dev)
    probs[0].C = Cm.data_ptr()
    probs[0].M, probs[0].N, probs[0].ldc = M, N, N + 3
    probs[0].seg_begin, probs[0].seg_end = 0, len(Ks)
    probs[0].alpha, probs[0].beta = 1.0, 0.0
    L.check(lib.cyb_gemm_grouped_f64(ctx, probs, 1, segs, len(Ks)))
    torch.cuda.synchronize()
    out = Cm.cpu().numpy()
    err = np.abs(out[:, :N] - ref).max() / max(1.0, np.abs(ref).max())
    pad_ok = np.all(out[:, N:] == 7.0)
    print(f'[gemm] M={M} N={N} Ks={Ks} ta={ta} tb={tb}: rel err {err:.2e} pad_untouched={pad_ok}')
    assert err < 1e-13 and pad_ok


def u1_leg(chi, sq):
    qmax = int(4 * sq)
    qs = np.arange(-qmax, qmax + 1)
    w = np.exp(-qs ** 2 / (2 * sq ** 2))
    m = np.floor(chi * w / w.sum()).astype(int)
    m[qs == 0] += chi - m.sum()
    keep = m > 0
    return qs[keep], m[keep]


def gemm_perf(chi):
    qs, ms = u1_leg(chi, 2.0)
    mult = dict(zip(qs.tolist(), ms.tolist()))
    # theta[(vL p0),(p1 vR)] = A[vL p0, c] B[c, p1 vR]; charges: qL + p0 = c ; c + p1 = qR
    plist = []
    for qL, mL in mult.items():
        for p0 in (-1, 1):
            c = qL + p0
            if c not in mult:
                continue
            for p1 in (-1, 1):
                qR = c + p1
                if qR not in mult:
                    continue
                plist.append((mL, mult[qR], mult[c]))
    n = len(plist)
    probs = (L.GemmProb * n)()
    segs = (L.GemmSeg * n)()
    keep = []
    flops = 0
    for i, (M, N, K) in enumerate(plist):
        A = torch.randn(M, K, dtype=torch.float64, device=dev)
        B = torch.randn(K, N, dtype=torch.float64, device=dev)
        Cm = torch.empty(M, N, dtype=torch.float64, device=dev)
        keep += [A, B, Cm]
        segs[i].A, segs[i].B, segs[i].K = A.data_ptr(), B.data_ptr(), K
        segs[i].a_rs, segs[i].a_cs, segs[i].b_rs, segs[i].b_cs = K, 1, N, 1
        probs[i].C, probs[i].M, probs[i].N, probs[i].ldc = Cm.data_ptr(), M, N, N
        probs[i].seg_begin, probs[i].seg_end, probs[i].alpha, probs[i].beta = i, i + 1, 1.0, 0.0
        flops += 2 * M * N * K
    plan = C.c_void_p()
    L.check(lib.cyb_gemm_plan_create(ctx, C.byref(plan), probs, n, segs, n))
    for _ in range(3):
        L.check(lib.cyb_gemm_plan_run(ctx, plan))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        L.check(lib.cyb_gemm_plan_run(ctx, plan))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # spot check the biggest problem
    i = int(np.argmax([m * n_ * k for m, n_, k in plist]))
    ref = keep[3 * i] @ keep[3 * i + 1]
    err = (keep[3 * i + 2] - ref).abs().max().item() / ref.abs().max().item()
    print(f'[gemm perf] chi={chi}: {n} GEMMs, {flops / 1e9:.2f} GFLOP, {ms * 1e3:.1f} us -> {flops / ms / 1e9:.2f} TFLOP/s; '
          f'dominant {plist[i]} rel err vs torch {err:.1e}')
    L.check(lib.cyb_gemm_plan_destroy(plan))


def svd_case(shapes, seed=0, rank=None, timing=False):
    rng = np.random.default_rng(seed)
    n = len(shapes)
    descs = (L.SvdDesc * n)()
    keep = []
    mats = []
    for i, (m, nn) in enumerate(shapes):
        a = rng.standard_normal((m, nn))
        if rank is not None:
            r = min(rank, m, nn)
            a = rng.standard_normal((m, r)) @ rng.standard_normal((r, nn))
        mats.append(a)
        k = min(m, nn)
        A = torch.from_numpy(a).to(dev)
        U = torch.empty(m, k, dtype=torch.float64, device=dev)
        S = torch.empty(k, dtype=torch.float64, device=dev)
        Vh = torch.empty(k, nn, dtype=torch.float64, device=dev)
        keep.append((A, U, S, Vh))
        descs[i].A, descs[i].lda, descs[i].m, descs[i].n = A.data_ptr(), nn, m, nn
        descs[i].U, descs[i].ldu, descs[i].S, descs[i].Vh, descs[i].ldvh = U.data_ptr(), k, S.data_ptr(), Vh.data_ptr(), nn
    info = (C.c_int32 * n)()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = lib.cyb_svd_batched_f64(ctx, descs, n, info)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if st != 0:
        print('[svd] status', st, lib.cyb_last_error().decode())
    worst = 0.0
    for i, (m, nn) in enumerate(shapes):
        A, U, S, Vh = [t.cpu().numpy() for t in keep[i]]
        sref = np.linalg.svd(mats[i], compute_uv=False)
        nrm = max(sref[0], 1e-300)
        e_s = np.abs(S - sref).max() / nrm
        e_r = np.abs((U * S) @ Vh - mats[i]).max() / nrm
        e_u = np.abs(U.T @ U - np.eye(U.shape[1])).max()
        e_v = np.abs(Vh @ Vh.T - np.eye(Vh.shape[0])).max()
        worst = max(worst, e_s, e_r, e_u, e_v)
        if not timing or max(e_s, e_r, e_u, e_v) > 1e-11:
            print(f'[svd] {m}x{nn} rank={rank}: sweeps={info[i]} dS={e_s:.1e} recon={e_r:.1e} UtU={e_u:.1e} VVt={e_v:.1e}')
    print(f'[svd] batch of {n}: {1e3 * (t1 - t0):.2f} ms wall, worst err {worst:.1e}, sweeps={list(info)}')
    return worst


def eigh_case(ns, seed=0):
    rng = np.random.default_rng(seed)
    n = len(ns)
    descs = (L.EighDesc * n)()
    keep, mats = [], []
    for i, k in enumerate(ns):
        a = rng.standard_normal((k, k))
        a = a + a.T
        mats.append(a)
        A = torch.from_numpy(a).to(dev)
        W = torch.empty(k, dtype=torch.float64, device=dev)
        V = torch.empty(k, k, dtype=torch.float64, device=dev)
        keep.append((A, W, V))
        descs[i].A, descs[i].lda, descs[i].n, descs[i].W, descs[i].V, descs[i].ldv = A.data_ptr(), k, k, W.data_ptr(), V.data_ptr(), k
    info = (C.c_int32 * n)()
    t0 = time.perf_counter()
    st = lib.cyb_eigh_batched_f64(ctx, descs, n, info)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if st != 0:
        print('[eigh] status', st, lib.cyb_last_error().decode())
    for i, k in enumerate(ns):
        A, W, V = [t.cpu().numpy() for t in keep[i]]
        wref = np.linalg.eigvalsh(mats[i])
        nrm = np.abs(wref).max()
        print(f'[eigh] n={k}: sweeps={info[i]} dW={np.abs(W - wref).max() / nrm:.1e} '
              f'resid={np.abs(mats[i] @ V - V * W).max() / nrm:.1e} VtV={np.abs(V.T @ V - np.eye(k)).max():.1e}')
    print(f'[eigh] batch wall {1e3 * (t1 - t0):.2f} ms')


def qr_case(shapes, full=False, seed=0):
    rng = np.random.default_rng(seed)
    n = len(shapes)
    descs = (L.QrDesc * n)()
    keep, mats = [], []
    for i, (m, nn) in enumerate(shapes):
        a = rng.standard_normal((m, nn))
        mats.append(a)
        k = min(m, nn)
        kq = m if full else k
        A = torch.from_numpy(a).to(dev)
        Q = torch.empty(m, kq, dtype=torch.float64, device=dev)
        R = torch.empty(kq, nn, dtype=torch.float64, device=dev)
        keep.append((A, Q, R))
        descs[i].A, descs[i].lda, descs[i].m, descs[i].n = A.data_ptr(), nn, m, nn
        descs[i].Q, descs[i].ldq, descs[i].R, descs[i].ldr, descs[i].full = Q.data_ptr(), kq, R.data_ptr(), nn, int(full)
    t0 = time.perf_counter()
    L.check(lib.cyb_qr_batched_f64(ctx, descs, n))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i, (m, nn) in enumerate(shapes):
        A, Q, R = [t.cpu().numpy() for t in keep[i]]
        import scipy.linalg
        qref, rref = scipy.linalg.qr(mats[i], mode='full' if full else 'economic')
        print(f'[qr] {m}x{nn} full={full}: recon={np.abs(Q @ R - mats[i]).max():.1e} QtQ={np.abs(Q.T @ Q - np.eye(Q.shape[1])).max():.1e} '
              f'tril={np.abs(np.tril(R, -1)).max():.1e} dR_vs_lapack={np.abs(R - rref).max():.1e}')
    print(f'[qr] batch wall {1e3 * (t1 - t0):.2f} ms')


if __name__ == '__main__':
    what = sys.argv[1:] or ['peak', 'gemm', 'gemmperf', 'svd', 'eigh', 'qr']
    print('device', torch.cuda.get_device_name(0))
    if 'peak' in what:
        peak()
    if 'gemm' in what:
        gemm_case(16, 16, [4])
        gemm_case(1, 1, [1])
        gemm_case(17, 5, [3])
        gemm_case(33, 47, [29])
        gemm_case(64, 64, [64])
        gemm_case(100, 90, [77, 13])
        gemm_case(130, 257, [65], ta=True)
        gemm_case(130, 257, [65], tb=True)
        gemm_case(200, 129, [31, 16, 1], ta=True, tb=True)
        gemm_case(824, 721, [824])
        gemm_case(300, 1, [50])
        gemm_case(1, 300, [50])
    if 'gemmperf' in what:
        gemm_perf(1024)
        gemm_perf(4096)
    if 'svd' in what:
        svd_case([(5, 3), (3, 5), (1, 1), (64, 64), (100, 37), (37, 100), (200, 300)])
        svd_case([(80, 80), (50, 120)], rank=3)
        svd_case([(360, 360), (335, 335), (246, 246)], timing=True)
    if 'svdbig' in what:
        qs, ms = u1_leg(4096, 2.0)
        shp = [(int(m) * 2 if False else int(m), int(m)) for m in ms]
        svd_case([(1442, 1442), (1236, 1236), (721, 721)], timing=True)
    if 'eigh' in what:
        eigh_case([1, 2, 7, 64, 100, 300])
    if 'qr' in what:
        qr_case([(5, 3), (3, 5), (64, 64), (300, 40), (40, 300)])
        qr_case([(5, 3), (3, 5), (70, 20)], full=True)
    print('FIRST LIGHT DONE')


def gemm_uniform(n, reps=10):
    A = torch.randn(n, n, dtype=torch.float64, device=dev)
    B = torch.randn(n, n, dtype=torch.float64, device=dev)
    Cm = torch.empty(n, n, dtype=torch.float64, device=dev)
    probs = (L.GemmProb * 1)()
    segs = (L.GemmSeg * 1)()
    segs[0].A, segs[0].B, segs[0].K = A.data_ptr(), B.data_ptr(), n
    segs[0].a_rs, segs[0].a_cs, segs[0].b_rs, segs[0].b_cs = n, 1, n, 1
    probs[0].C, probs[0].M, probs[0].N, probs[0].ldc = Cm.data_ptr(), n, n, n
    probs[0].seg_begin, probs[0].seg_end, probs[0].alpha, probs[0].beta = 0, 1, 1.0, 0.0
    plan = C.c_void_p()
    L.check(lib.cyb_gemm_plan_create(ctx, C.byref(plan), probs, 1, segs, 1))
    for _ in range(2):
        L.check(lib.cyb_gemm_plan_run(ctx, plan))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.cyb_gemm_plan_run(ctx, plan))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'[gemm uniform] {n}^3: {ms:.3f} ms -> {2 * n ** 3 / ms / 1e9:.2f} TFLOP/s')
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        torch.matmul(A, B, out=Cm)
    t1.record()
    torch.cuda.synchronize()
    ms2 = t0.elapsed_time(t1) / reps
    print(f'[gemm uniform] {n}^3 torch.matmul (rocBLAS/hipBLASLt, comparison only): {ms2:.3f} ms -> {2 * n ** 3 / ms2 / 1e9:.2f} TFLOP/s')
    L.check(lib.cyb_gemm_plan_destroy(plan))


if 'gemmuniform' in sys.argv[1:]:
    gemm_uniform(2048)
    gemm_uniform(4096)
    gemm_uniform(8192, reps=3)


def gemm_list(shapes, reps=20, label=''):
    n = len(shapes)
    probs = (L.GemmProb * n)()
    segs = (L.GemmSeg * n)()
    keep = []
    flops = 0
    for i, (M, N, K) in enumerate(shapes):
        A = torch.randn(M, K, dtype=torch.float64, device=dev)
        B = torch.randn(K, N, dtype=torch.float64, device=dev)
        Cm = torch.empty(M, N, dtype=torch.float64, device=dev)
        keep += [A, B, Cm]
        segs[i].A, segs[i].B, segs[i].K = A.data_ptr(), B.data_ptr(), K
        segs[i].a_rs, segs[i].a_cs, segs[i].b_rs, segs[i].b_cs = K, 1, N, 1
        probs[i].C, probs[i].M, probs[i].N, probs[i].ldc = Cm.data_ptr(), M, N, N
        probs[i].seg_begin, probs[i].seg_end, probs[i].alpha, probs[i].beta = i, i + 1, 1.0, 0.0
        flops += 2 * M * N * K
    plan = C.c_void_p()
    L.check(lib.cyb_gemm_plan_create(ctx, C.byref(plan), probs, n, segs, n))
    for _ in range(3):
        L.check(lib.cyb_gemm_plan_run(ctx, plan))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.cyb_gemm_plan_run(ctx, plan))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'[gemm list] {label}: {n} GEMMs {flops / 1e9:.2f} GFLOP {ms * 1e3:.1f} us -> {flops / ms / 1e9:.2f} TFLOP/s')
    L.check(lib.cyb_gemm_plan_destroy(plan))


if 'gemmlist' in sys.argv[1:]:
    gemm_list([(824, 721, 824)], label='dominant alone (42 tiles)')
    gemm_list([(824, 720, 824)], label='dominant, N even')
    gemm_list([(768, 640, 832)], label='no edge tiles (30 tiles)')
    gemm_list([(824, 721, 824)] * 6, label='6x dominant (252 tiles)')
    gemm_list([(824, 721, 824)] * 12, label='12x dominant (504 tiles)')
    gemm_list([(824, 721, 824)] * 24, label='24x dominant (1008 tiles)')
    gemm_list([(1024, 1024, 1024)] * 8, label='8x 1024^3 (512 tiles)')
