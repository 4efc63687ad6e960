// complex128 QR of small blocks: one workgroup per block, classical Gram-Schmidt with reorthogonalisation in LDS.
//
// scipy.linalg.qr(a, mode='economic' | 'full') on complex128 blocks (NumpyBlockBackend::matrix_qr, numpy.cpp:1236-1245;
// matrix_lq is the same on the transposed view, block_backend.cpp:1033-1040).  The columns are processed in order; every
// column is projected twice against the finished q's (CGS2: orthogonality at rounding level for numerically independent
// columns), a column that is numerically dependent gets a unit vector orthogonalised against the others as its q and a
// zero on the diagonal of R (Q stays unitary and R upper triangular, as with Householder QR), 'full' completes Q to m
// columns the same way.  Limits: m <= 128 and 16 B * kq * (m | 1) <= 150 KB (kq = m in 'full' mode, else min(m, n));
// larger blocks run the blocked version of the same scheme in csvd_large.hip.
#include "common.h"

#include <algorithm>
#include <vector>

namespace cyb_clarge {
struct QrReq {
    const double* A;
    double *Q, *R;
    int64_t lda, ldq, ldr;
    int32_t m, n, kq;
};
int run_qr(cyb_ctx_t ctx, const std::vector<QrReq>& req);
} // namespace cyb_clarge

namespace {

constexpr int NT = 256;
constexpr int MAXM = 128;
constexpr size_t LDS_BUDGET = 150 * 1024;

typedef double d2 __attribute__((ext_vector_type(2)));

struct QDesc {
    const double* A;
    double *Q, *R;
    int64_t lda, ldq, ldr;
    int32_t m, n, kq;
};

__device__ __forceinline__ d2 cmul(d2 a, d2 b) { return d2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ double g8(double v)
{
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}
__device__ double bsum(double v, double* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(NT) cqr_small_kernel(const QDesc* __restrict__ descs)
{
    extern __shared__ __attribute__((aligned(16))) double smem_raw[];
    __shared__ double red[NT / 64];
    __shared__ d2 col[MAXM];
    __shared__ d2 coef[MAXM];
    __shared__ d2 racc[MAXM];
    d2* Qs = reinterpret_cast<d2*>(smem_raw); // Qs[c * ldm + r]
    const QDesc d = descs[blockIdx.x];
    const int tid = threadIdx.x, grp = tid >> 3, l8 = tid & 7;
    const int m = d.m, n = d.n, kq = d.kq, ldm = m | 1;
    const d2* A = reinterpret_cast<const d2*>(d.A);
    d2* Q = reinterpret_cast<d2*>(d.Q);
    d2* R = reinterpret_cast<d2*>(d.R);
    // R = 0 (its strictly lower part and the rows below n of 'full' mode stay zero)
    for (int e = tid; e < kq * n; e += NT) R[(int64_t)(e / n) * d.ldr + (e % n)] = d2{0.0, 0.0};
    // scale: ||A||_F for the dependence threshold
    double f2 = 0.0;
    for (int e = tid; e < m * n; e += NT) {
        const d2 v = A[(int64_t)(e / n) * d.lda + (e % n)];
        f2 += v.x * v.x + v.y * v.y;
    }
    f2 = bsum(f2, red);
    const double thresh = sqrt(f2) * 2.3e-16 * m;
    int nq = 0; // finished columns of Q
    // orthogonalise `col` (LDS, also in register `a` of thread tid < m) against q_0..q_{nq-1}, twice; racc = coefficients
    auto project = [&](d2& a) {
        for (int i = tid; i < nq; i += NT) racc[i] = d2{0.0, 0.0};
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            for (int i = grp; i < nq; i += NT / 8) {
                double dr = 0.0, di = 0.0;
                for (int r = l8; r < m; r += 8) {
                    const d2 q = Qs[i * ldm + r], x = col[r];
                    dr += q.x * x.x + q.y * x.y; // conj(q) * x
                    di += q.x * x.y - q.y * x.x;
                }
                dr = g8(dr);
                di = g8(di);
                if (l8 == 0) coef[i] = d2{dr, di};
            }
            __syncthreads();
            if (tid < m) {
                for (int i = 0; i < nq; ++i) a -= cmul(coef[i], Qs[i * ldm + tid]);
                col[tid] = a;
            }
            for (int i = tid; i < nq; i += NT) racc[i] += coef[i];
        }
        __syncthreads();
    };
    // a unit vector orthogonalised against the finished columns becomes column nq of Q
    auto complete = [&](int start) {
        for (int cand = 0; cand < m; ++cand) {
            const int e = (start + cand) % m;
            d2 a = d2{(tid == e) ? 1.0 : 0.0, 0.0};
            __syncthreads();
            if (tid < m) col[tid] = a;
            project(a);
            const double n2 = bsum(tid < m ? a.x * a.x + a.y * a.y : 0.0, red);
            if (n2 * m > 0.5) { // (uniform) some unit vector has a residual of at least 1 / m
                if (tid < m) Qs[nq * ldm + tid] = a * (1.0 / sqrt(n2));
                __syncthreads();
                return;
            }
        }
    };
    for (int j = 0; j < n; ++j) {
        d2 a = d2{0.0, 0.0};
        if (tid < m) {
            a = A[(int64_t)tid * d.lda + j];
            col[tid] = a;
        }
        project(a);
        for (int i = tid; i < nq; i += NT) R[(int64_t)i * d.ldr + j] = racc[i];
        if (j < kq) { // this column defines q_j (nq == j here)
            const double n2 = bsum(tid < m ? a.x * a.x + a.y * a.y : 0.0, red);
            const double nrm = sqrt(n2);
            if (nrm > thresh) {
                if (tid < m) Qs[nq * ldm + tid] = a * (1.0 / nrm);
                if (tid == 0) R[(int64_t)j * d.ldr + j] = d2{nrm, 0.0};
                __syncthreads();
            } else {
                complete(j); // dependent column: R[j][j] stays 0
            }
            ++nq;
        }
    }
    while (nq < kq) { // 'full' mode with m > n (or kq > n): complete Q
        complete(nq);
        ++nq;
    }
    __syncthreads();
    for (int e = tid; e < m * kq; e += NT) {
        const int r = e / kq, c = e - r * kq;
        Q[(int64_t)r * d.ldq + c] = Qs[c * ldm + r];
    }
}

size_t lds_bytes(int64_t m, int64_t kq) { return sizeof(double) * 2 * (size_t)kq * (size_t)(m | 1); }

} // namespace

extern "C" int cyb_qr_batched_c128(cyb_ctx_t ctx, const cyb_qr_desc* descs, int64_t n)
{
    CYB_REQUIRE(ctx, "cyb_qr_batched_c128: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_qr_batched_c128: bad descriptor list");
    std::vector<QDesc> hd;
    std::vector<cyb_clarge::QrReq> big;
    size_t lds = 0;
    for (int64_t i = 0; i < n; ++i) {
        const cyb_qr_desc& s = descs[i];
        CYB_REQUIRE(s.m >= 0 && s.n >= 0, "qr block %lld: negative extent", (long long)i);
        const int64_t kq = s.full ? s.m : std::min(s.m, s.n);
        if (s.m == 0 || kq == 0) continue;
        CYB_REQUIRE(s.A || s.n == 0, "qr block %lld: A is NULL", (long long)i);
        CYB_REQUIRE(s.Q && (s.R || s.n == 0), "qr block %lld: NULL output", (long long)i);
        CYB_REQUIRE(s.lda >= s.n && s.ldq >= kq && s.ldr >= s.n, "qr block %lld: leading dimension too small", (long long)i);
        if (s.m > MAXM || s.n > 4 * MAXM || lds_bytes(s.m, kq) > LDS_BUDGET) { // beyond the in-LDS limit: csvd_large.hip
            CYB_REQUIRE(s.m < (1 << 30) && s.n < (1 << 30), "qr block %lld: extent too large", (long long)i);
            big.push_back(cyb_clarge::QrReq{s.A, s.Q, s.R, s.lda, s.ldq, s.ldr, (int32_t)s.m, (int32_t)s.n, (int32_t)kq});
            continue;
        }
        hd.push_back(QDesc{s.A, s.Q, s.R, s.lda, s.ldq, s.ldr, (int32_t)s.m, (int32_t)s.n, (int32_t)kq});
        lds = std::max(lds, lds_bytes(s.m, kq));
    }
    if (!big.empty()) CYB_TRY(cyb_clarge::run_qr(ctx, big));
    if (hd.empty()) return CYB_OK;
    static bool attr_set = false;
    if (!attr_set) {
        CYB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cqr_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LDS_BUDGET));
        attr_set = true;
    }
    void* d_descs = nullptr;
    CYB_TRY(ctx->upload(hd.data(), sizeof(QDesc) * hd.size(), &d_descs));
    hipLaunchKernelGGL(cqr_small_kernel, dim3((unsigned)hd.size()), dim3(NT), lds, ctx->stream, static_cast<const QDesc*>(d_descs));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}
