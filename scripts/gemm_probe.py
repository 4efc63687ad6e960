"""Run ONE uniform GEMM plan a few times (for rocprofv3 --pmc passes)."""
import sys
sys.argv = [sys.argv[0], 'none'] + sys.argv[1:]
exec(open('scripts/first_light.py').read().split("if __name__ == '__main__':")[0])
torch.manual_seed(0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
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
print('probe done')
