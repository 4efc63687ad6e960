"""Latency of the two GEMM shapes of a blocked-QR panel step (development aid): g1 = W(32 x nt) = V^T A with K = rows
in two chunks, g3 = A(nt x mr) -= W^T V^T as two K=32 segments, for the chi=4096 block sizes."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from cyten_amd import _lib as L
from cyten_amd.runtime import get_context
ctxo = get_context(0); lib, ctx = ctxo.lib, ctxo.handle
dev = torch.device('cuda:0')
sizes = [1442, 1236, 1236, 824, 824, 418, 418, 150, 150]
def run(build, reps=50, label=''):
    probs, segs, keep = build()
    n, ns = len(probs), len(segs)
    P = (L.GemmProb * n)(*probs); S = (L.GemmSeg * ns)(*segs)
    for _ in range(5):
        L.check(lib.cyb_gemm_grouped_enqueue_f64(ctx, P, n, S, ns))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.cyb_gemm_grouped_enqueue_f64(ctx, P, n, S, ns))
    e1.record(); torch.cuda.synchronize()
    print(f'{label}: {1e3 * e0.elapsed_time(e1) / reps:.1f} us per launch ({n} problems, {ns} segments)')
def mk(seg_k, nseg, beta, j0=64):
    def build():
        probs, segs, keep = [], [], []
        for m in sizes:
            mr, nt = m - j0, m - j0 - 32
            A = torch.randn(nt, mr, dtype=torch.float64, device=dev)
            V = torch.randn(mr, seg_k * nseg, dtype=torch.float64, device=dev)
            W = torch.randn(seg_k * nseg, nt, dtype=torch.float64, device=dev)
            keep += [A, V, W]
            s0 = len(segs)
            for s in range(nseg):
                g = L.GemmSeg(); g.A, g.B, g.K = W.data_ptr() + 8 * s * seg_k * nt, V.data_ptr() + 8 * s * seg_k, seg_k
                g.a_rs, g.a_cs, g.b_rs, g.b_cs = 1, nt, 1, seg_k * nseg     # A^T-like views as in bqr_factor
                segs.append(g)
            p = L.GemmProb(); p.C, p.M, p.N, p.ldc = A.data_ptr(), nt, mr, mr
            p.seg_begin, p.seg_end, p.alpha, p.beta = s0, len(segs), -1.0, beta
            probs.append(p)
        return probs, segs, keep
    return build
run(mk(32, 2, 1.0), label='g3: K = 2 x 32, beta = 1')
run(mk(64, 1, 1.0), label='g3: K = 64 in one segment, beta = 1')
run(mk(32, 1, 1.0), label='g3: K = 32 in one segment, beta = 1')
run(mk(32, 2, 0.0), label='g3: K = 2 x 32, beta = 0')
run(mk(16, 1, 1.0), label='g3: K = 16, beta = 1')
def g1():
    probs, segs, keep = [], [], []
    for m in sizes:
        mr, nt = m - 64, m - 96
        A = torch.randn(nt, mr, dtype=torch.float64, device=dev); V = torch.randn(32, mr, dtype=torch.float64, device=dev)
        for s in range(2):
            rows = mr // 2
            Wp = torch.empty(32, nt, dtype=torch.float64, device=dev); keep += [Wp]
            g = L.GemmSeg(); g.A, g.B, g.K = V.data_ptr() + 8 * s * rows, A.data_ptr() + 8 * s * rows, rows
            g.a_rs, g.a_cs, g.b_rs, g.b_cs = mr, 1, 1, mr
            segs.append(g)
            p = L.GemmProb(); p.C, p.M, p.N, p.ldc = Wp.data_ptr(), 32, nt, nt
            p.seg_begin, p.seg_end, p.alpha, p.beta = len(segs) - 1, len(segs), 1.0, 0.0
            probs.append(p)
        keep += [A, V]
    return probs, segs, keep
run(g1, label='g1: 32 x nt, K = rows / 2, two chunks per block')
