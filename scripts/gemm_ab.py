"""GEMM timing for A/B runs: uniform n^3 and the chi=4096 theta list (run with CYTEN_AMD_LIB=... for variants)."""
import sys
sys.argv = [sys.argv[0], 'none'] + sys.argv[1:]
exec(open('scripts/first_light.py').read().split("if __name__ == '__main__':")[0])
torch.manual_seed(0)
def uniform(n, reps=10):
    A = torch.randn(n, n, dtype=torch.float64, device=dev)
    B = torch.randn(n, n, dtype=torch.float64, device=dev)
    Cm = torch.empty(n, n, dtype=torch.float64, device=dev)
    probs = (L.GemmProb * 1)(); segs = (L.GemmSeg * 1)()
    segs[0].A, segs[0].B, segs[0].K = A.data_ptr(), B.data_ptr(), n
    segs[0].a_rs, segs[0].a_cs, segs[0].b_rs, segs[0].b_cs = n, 1, n, 1
    probs[0].C, probs[0].M, probs[0].N, probs[0].ldc = Cm.data_ptr(), n, n, n
    probs[0].seg_begin, probs[0].seg_end, probs[0].alpha, probs[0].beta = 0, 1, 1.0, 0.0
    plan = C.c_void_p()
    L.check(lib.cyb_gemm_plan_create(ctx, C.byref(plan), probs, 1, segs, 1))
    for _ in range(3):
        L.check(lib.cyb_gemm_plan_run(ctx, plan))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.cyb_gemm_plan_run(ctx, plan))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    err = (Cm - A @ B).abs().max().item()
    print(f'[uniform] n={n}: {ms*1e3:.1f} us -> {2*n**3/ms/1e9:.2f} TFLOP/s  err {err:.1e}')
    L.check(lib.cyb_gemm_plan_destroy(plan))
for n in (4096, 2048):
    uniform(n)
gemm_perf(4096)
gemm_perf(2048)
gemm_perf(1024)
