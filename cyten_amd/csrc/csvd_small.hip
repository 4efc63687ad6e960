// complex128 SVD and Hermitian eigh of small blocks: one workgroup per block, complex one-sided Jacobi in LDS.
//
// The complex counterpart of svd_small.hip for the reference's second dtype (NumpyBlockBackend::matrix_svd
// numpy.cpp:1247-1297 and ::eigh :658-680 on complex128 blocks; the reference's own test-suite only has small blocks).
// A pair of complex columns (a_p, a_q) with g = a_p^H a_q is orthogonalised by rotating a_q with the phase e^{-i arg g}
// (which makes the inner product real and non-negative) followed by the real Jacobi rotation; the same 2x2 unitary is
// applied to the columns of V, so W = A V stays true and at convergence A = (W / sigma) diag(sigma) V^H.
// eigh: the Hermitian block is shifted by s = ||H||_F to a positive semi-definite one, whose right singular vectors ARE
// the eigenvectors (any basis of a degenerate eigenspace is one), lambda = sigma - s, returned ascending.
// Larger complex blocks go to the device-memory Jacobi of csvd_large.hip (functional path, not MFMA-blocked).
#include "common.h"

#include <algorithm>
#include <vector>

namespace cyb_clarge {
struct Req {
    const double* A;
    double *U, *S, *Vh;
    int64_t lda, ldu, ldvh;
    int32_t m, n, mode;
};
int run(cyb_ctx_t ctx, const std::vector<Req>& req, int32_t* sweeps_out);
} // namespace cyb_clarge

namespace {

constexpr int NT = 256;
constexpr int MAXN = 64, MAXM = 128;
constexpr int MAX_SWEEPS = 40;
constexpr size_t LDS_BUDGET = 150 * 1024;

typedef double d2 __attribute__((ext_vector_type(2))); // (re, im)

struct CDesc {
    const double* A;          // complex interleaved, lda in complex elements
    double *U, *S, *Vh;       // SVD: U, Vh complex, S real.  eigh: S = W (real, ascending), U = V (complex), Vh unused
    int64_t lda, ldu, ldvh;
    int32_t m, n, mode;       // mode 0: SVD, 1: eigh
};

__device__ __forceinline__ d2 cmul(d2 a, d2 b) { return d2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ d2 cmulc(d2 a, d2 b) { return d2{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x}; } // conj(a) * b
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

__global__ void __launch_bounds__(NT) csvd_small_kernel(const CDesc* __restrict__ descs, int32_t* __restrict__ sweeps_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem_raw[];
    __shared__ double red[NT / 64];
    __shared__ double sig[MAXN];
    __shared__ int rank_of[MAXN];
    __shared__ d2 dots[MAXN];
    __shared__ int flag;
    d2* smem = reinterpret_cast<d2*>(smem_raw);
    const CDesc d = descs[blockIdx.x];
    const int tid = threadIdx.x;
    const bool tall = d.m >= d.n;
    const int M = tall ? d.m : d.n;
    const int N = tall ? d.n : d.m;
    const int Np = (N + 1) & ~1;
    const int ldw = M | 1, ldv = Np | 1;
    d2* W = smem;
    d2* V = smem + (size_t)Np * ldw;
    const d2* A = reinterpret_cast<const d2*>(d.A);
    // ---- load.  A wide block is worked on as its conjugate transpose: A^H = U' S V'^H  =>  A = V' S U'^H.
    for (int e = tid; e < Np * M; e += NT) {
        const int c = e / M, r = e - c * M;
        d2 v = d2{0.0, 0.0};
        if (c < N) {
            if (tall) v = A[(int64_t)r * d.lda + c];
            else {
                v = A[(int64_t)c * d.lda + r];
                v.y = -v.y;
            }
        }
        W[c * ldw + r] = v;
    }
    for (int e = tid; e < Np * Np; e += NT) {
        const int c = e / Np, r = e - c * Np;
        V[c * ldv + r] = d2{(c == r) ? 1.0 : 0.0, 0.0};
    }
    __syncthreads();
    // range safety: scale the block by a power of two so that its largest entry is of order one (the squares of
    // entries beyond 1e+-154 would leave the double range); singular values / eigenvalues are scaled back at the end
    double amax = 0.0;
    for (int e = tid; e < Np * M; e += NT) {
        const int c = e / M, r = e - c * M;
        const d2 v = W[c * ldw + r];
        amax = fmax(amax, fmax(fabs(v.x), fabs(v.y)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    const double scl = (amax > 0.0 && amax < __builtin_huge_val()) ? scalbn(1.0, -ilogb(amax)) : 1.0;
    const double unscl = 1.0 / scl;
    if (scl != 1.0) {
        for (int e = tid; e < Np * M; e += NT) {
            const int c = e / M, r = e - c * M;
            W[c * ldw + r] *= scl;
        }
        __syncthreads();
    }
    double fro2 = 0.0;
    for (int e = tid; e < Np * M; e += NT) {
        const int c = e / M, r = e - c * M;
        const d2 v = W[c * ldw + r];
        fro2 += v.x * v.x + v.y * v.y;
    }
    fro2 = bsum(fro2, red);
    double shift = 0.0;
    if (d.mode == 1) { // eigh: H + ||H||_F I is positive semi-definite
        shift = sqrt(fro2);
        __syncthreads();
        for (int c = tid; c < N; c += NT) W[c * ldw + c].x += shift;
        __syncthreads();
        fro2 += shift * shift * N + 2.0 * shift * 0.0; // (bound only: used for the null threshold)
    }
    const double null2 = fro2 * (2.3e-16 * M) * (2.3e-16 * M);
    const int grp = tid >> 3, l8 = tid & 7;
    const int npairs = Np / 2;
    const double tol = 1e-15;
    int sweeps = 0;
    bool converged = (Np < 2);
    while (!converged && sweeps < MAX_SWEEPS) {
        double off = 0.0;
        for (int r = 0; r < Np - 1; ++r) {
            for (int k = grp; k < npairs; k += NT / 8) {
                const int p = (k == 0) ? Np - 1 : (r + k) % (Np - 1);
                const int q = (k == 0) ? r : (r - k + Np - 1) % (Np - 1);
                d2* wp = W + p * ldw;
                d2* wq = W + q * ldw;
                double app = 0.0, aqq = 0.0, gr = 0.0, gi = 0.0;
                for (int i = l8; i < M; i += 8) {
                    const d2 x = wp[i], y = wq[i];
                    app += x.x * x.x + x.y * x.y;
                    aqq += y.x * y.x + y.y * y.y;
                    gr += x.x * y.x + x.y * y.y; // conj(x) * y
                    gi += x.x * y.y - x.y * y.x;
                }
                app = g8(app);
                aqq = g8(aqq);
                gr = g8(gr);
                gi = g8(gi);
                const double ag = sqrt(gr * gr + gi * gi);
                const double den = sqrt(app) * sqrt(aqq);
                if (app > null2 && aqq > null2 && ag > tol * den) {
                    off = fmax(off, ag / den);
                    const d2 ph = d2{gr / ag, -gi / ag}; // e^{-i arg g}
                    const double zeta = (aqq - app) / (2.0 * ag);
                    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    for (int i = l8; i < M; i += 8) {
                        const d2 x = wp[i], y = cmul(wq[i], ph);
                        wp[i] = c * x - s * y;
                        wq[i] = s * x + c * y;
                    }
                    d2* vp = V + p * ldv;
                    d2* vq = V + q * ldv;
                    for (int i = l8; i < Np; i += 8) {
                        const d2 x = vp[i], y = cmul(vq[i], ph);
                        vp[i] = c * x - s * y;
                        vq[i] = s * x + c * y;
                    }
                }
            }
            __syncthreads();
        }
        ++sweeps;
        const unsigned long long any = __ballot(off > 0.0);
        if (tid == 0) flag = 0;
        __syncthreads();
        if ((tid & 63) == 0 && any) flag = 1;
        __syncthreads();
        converged = (flag == 0);
        __syncthreads();
    }
    if (tid == 0) sweeps_out[blockIdx.x] = converged ? sweeps : -1;
    // ---- singular values and their descending order
    for (int c = grp; c < Np; c += NT / 8) {
        double s2 = 0.0;
        for (int i = l8; i < M; i += 8) {
            const d2 v = W[c * ldw + i];
            s2 += v.x * v.x + v.y * v.y;
        }
        s2 = g8(s2);
        if (l8 == 0) sig[c] = (c < N) ? sqrt(s2) : -1.0;
    }
    __syncthreads();
    if (tid < Np) {
        const double s = sig[tid];
        int rk = 0;
        for (int c = 0; c < Np; ++c) rk += (sig[c] > s) || (sig[c] == s && c < tid);
        rank_of[tid] = rk;
    }
    __syncthreads();
    if (d.mode == 1) { // eigenpairs, ascending: lambda = sigma - shift, eigenvectors = columns of V
        double* Wout = d.S;
        d2* Vout = reinterpret_cast<d2*>(d.U);
        for (int c = tid; c < N; c += NT) Wout[N - 1 - rank_of[c]] = (sig[c] - shift) * unscl;
        for (int e = tid; e < N * N; e += NT) {
            const int j = e / N, c = e - j * N;
            Vout[(int64_t)j * d.ldu + (N - 1 - rank_of[c])] = V[c * ldv + j];
        }
        return;
    }
    const double thresh = sqrt(null2);
    for (int c = grp; c < N; c += NT / 8) {
        if (sig[c] > thresh) {
            const double inv = 1.0 / sig[c];
            for (int i = l8; i < M; i += 8) W[c * ldw + i] *= inv;
        }
    }
    __syncthreads();
    // ---- null directions: unit vectors orthogonalised against the finished columns (twice)
    for (int c = 0; c < N; ++c) {
        if (sig[c] > thresh) continue;
        bool done = false;
        for (int cand = 0; cand < M && !done; ++cand) {
            const int e = (cand + c) % M;
            d2 v = d2{0.0, 0.0};
            if (tid < M) {
                v = d2{(tid == e) ? 1.0 : 0.0, 0.0};
                for (int k = 0; k < N; ++k)
                    if (k != c && (sig[k] > thresh || k < c)) {
                        const d2 qe = W[k * ldw + e]; // <q_k, e_e> = conj(q_k[e])
                        v -= cmul(d2{qe.x, -qe.y}, W[k * ldw + tid]);
                    }
            }
            const double n2 = bsum(v.x * v.x + v.y * v.y, red);
            if (n2 * M > 0.5) {
                __syncthreads();
                if (tid < M) W[c * ldw + tid] = v;
                __syncthreads();
                for (int k = grp; k < N; k += NT / 8) {
                    double dr = 0.0, di = 0.0;
                    if (k != c && (sig[k] > thresh || k < c))
                        for (int i = l8; i < M; i += 8) {
                            const d2 t = cmulc(W[k * ldw + i], W[c * ldw + i]);
                            dr += t.x;
                            di += t.y;
                        }
                    dr = g8(dr);
                    di = g8(di);
                    if (l8 == 0) dots[k] = d2{dr, di};
                }
                __syncthreads();
                if (tid < M) {
                    for (int k = 0; k < N; ++k)
                        if (k != c && (sig[k] > thresh || k < c)) v -= cmul(dots[k], W[k * ldw + tid]);
                }
                const double n3 = bsum(v.x * v.x + v.y * v.y, red);
                if (tid < M) W[c * ldw + tid] = v * (1.0 / sqrt(n3));
                __syncthreads();
                done = true;
            }
        }
    }
    __syncthreads();
    // ---- write U (m x k), S (k), Vh (k x n)
    d2* U = reinterpret_cast<d2*>(d.U);
    d2* Vh = reinterpret_cast<d2*>(d.Vh);
    for (int c = tid; c < N; c += NT) d.S[rank_of[c]] = fmax(sig[c], 0.0) * unscl;
    if (tall) { // A = (W / sigma) S V^H
        for (int e = tid; e < N * M; e += NT) {
            const int r = e / N, c = e - r * N;
            U[(int64_t)r * d.ldu + rank_of[c]] = W[c * ldw + r];
        }
        for (int e = tid; e < N * N; e += NT) {
            const int c = e / N, j = e - c * N;
            const d2 v = V[c * ldv + j];
            Vh[(int64_t)rank_of[c] * d.ldvh + j] = d2{v.x, -v.y};
        }
    } else { // A^H = (W / sigma) S V^H  =>  A = V S (W / sigma)^H
        for (int e = tid; e < N * N; e += NT) {
            const int j = e / N, c = e - j * N;
            U[(int64_t)j * d.ldu + rank_of[c]] = V[c * ldv + j];
        }
        for (int e = tid; e < N * M; e += NT) {
            const int c = e / M, r = e - c * M;
            const d2 v = W[c * ldw + r];
            Vh[(int64_t)rank_of[c] * d.ldvh + r] = d2{v.x, -v.y};
        }
    }
}

size_t lds_bytes(int64_t m, int64_t n)
{
    const int64_t M = std::max(m, n), N = std::min(m, n), Np = (N + 1) & ~(int64_t)1;
    return sizeof(double) * 2 * ((size_t)Np * (size_t)(M | 1) + (size_t)Np * (size_t)(Np | 1));
}

bool fits(int64_t m, int64_t n)
{
    return std::min(m, n) >= 1 && std::min(m, n) <= MAXN && std::max(m, n) <= MAXM && lds_bytes(m, n) <= LDS_BUDGET;
}

int launch(cyb_ctx_t ctx, std::vector<CDesc>& hd, size_t lds, int32_t* info)
{
    const int64_t n = (int64_t)hd.size();
    static bool attr_set = false;
    if (!attr_set) {
        CYB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(csvd_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LDS_BUDGET));
        attr_set = true;
    }
    void *d_descs = nullptr, *d_sw = nullptr;
    CYB_TRY(ctx->upload(hd.data(), sizeof(CDesc) * hd.size(), &d_descs));
    CYB_TRY(ctx->workspace(sizeof(int32_t) * (size_t)n, &d_sw, 3));
    hipLaunchKernelGGL(csvd_small_kernel, dim3((unsigned)n), dim3(NT), lds, ctx->stream, static_cast<const CDesc*>(d_descs),
                       static_cast<int32_t*>(d_sw));
    CYB_HIP(hipGetLastError());
    std::vector<int32_t> sw((size_t)n);
    CYB_HIP(hipMemcpyAsync(sw.data(), d_sw, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    CYB_HIP(hipStreamSynchronize(ctx->stream));
    int st = CYB_OK;
    for (int64_t i = 0; i < n; ++i) {
        if (info) info[i] = sw[(size_t)i];
        if (sw[(size_t)i] < 0) st = CYB_ERR_NOCONV;
    }
    if (st == CYB_ERR_NOCONV) cyb::set_error("complex small-block Jacobi: a block did not converge in %d sweeps", MAX_SWEEPS);
    return st;
}

// the in-LDS blocks in one launch, the others through the device-memory path
int finish(cyb_ctx_t ctx, std::vector<CDesc>& hd, const std::vector<int64_t>& idx, size_t lds, const std::vector<cyb_clarge::Req>& big,
           const std::vector<int64_t>& big_idx, int32_t* info)
{
    if (!hd.empty()) {
        std::vector<int32_t> inf(hd.size());
        const int st = launch(ctx, hd, lds, inf.data());
        if (info)
            for (size_t k = 0; k < idx.size(); ++k) info[idx[k]] = inf[k];
        if (st != CYB_OK) return st;
    }
    if (!big.empty()) {
        std::vector<int32_t> inf(big.size());
        const int st = cyb_clarge::run(ctx, big, inf.data());
        if (info)
            for (size_t k = 0; k < big_idx.size(); ++k) info[big_idx[k]] = inf[k];
        if (st != CYB_OK) return st;
    }
    return CYB_OK;
}

} // namespace

extern "C" {

int cyb_svd_batched_c128(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* info)
{
    CYB_REQUIRE(ctx, "cyb_svd_batched_c128: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_svd_batched_c128: bad descriptor list");
    std::vector<CDesc> hd;
    std::vector<int64_t> idx, big_idx;
    std::vector<cyb_clarge::Req> big;
    size_t lds = 0;
    for (int64_t i = 0; i < n; ++i) {
        const cyb_svd_desc& s = descs[i];
        if (info) info[i] = 0;
        CYB_REQUIRE(s.m >= 0 && s.n >= 0, "svd block %lld: negative extent", (long long)i);
        if (s.m == 0 || s.n == 0) continue;
        CYB_REQUIRE(s.A && s.U && s.S && s.Vh, "svd block %lld: NULL pointer", (long long)i);
        CYB_REQUIRE(s.lda >= s.n && s.ldu >= std::min(s.m, s.n) && s.ldvh >= s.n, "svd block %lld: leading dimension too small",
                    (long long)i);
        if (!fits(s.m, s.n)) {
            CYB_REQUIRE(s.m < (1 << 30) && s.n < (1 << 30), "svd block %lld: extent too large", (long long)i);
            big.push_back(cyb_clarge::Req{s.A, s.U, s.S, s.Vh, s.lda, s.ldu, s.ldvh, (int32_t)s.m, (int32_t)s.n, 0});
            big_idx.push_back(i);
            continue;
        }
        hd.push_back(CDesc{s.A, s.U, s.S, s.Vh, s.lda, s.ldu, s.ldvh, (int32_t)s.m, (int32_t)s.n, 0});
        idx.push_back(i);
        lds = std::max(lds, lds_bytes(s.m, s.n));
    }
    return finish(ctx, hd, idx, lds, big, big_idx, info);
}

int cyb_eigh_batched_c128(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* info)
{
    CYB_REQUIRE(ctx, "cyb_eigh_batched_c128: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_eigh_batched_c128: bad descriptor list");
    std::vector<CDesc> hd;
    std::vector<int64_t> idx, big_idx;
    std::vector<cyb_clarge::Req> big;
    size_t lds = 0;
    for (int64_t i = 0; i < n; ++i) {
        const cyb_eigh_desc& s = descs[i];
        if (info) info[i] = 0;
        CYB_REQUIRE(s.n >= 0, "eigh block %lld: negative extent", (long long)i);
        if (s.n == 0) continue;
        CYB_REQUIRE(s.A && s.W && s.V, "eigh block %lld: NULL pointer (eigenvectors are always computed for complex blocks)", (long long)i);
        CYB_REQUIRE(s.lda >= s.n && s.ldv >= s.n, "eigh block %lld: leading dimension too small", (long long)i);
        if (!fits(s.n, s.n)) {
            CYB_REQUIRE(s.n < (1 << 30), "eigh block %lld: extent too large", (long long)i);
            big.push_back(cyb_clarge::Req{s.A, s.V, s.W, nullptr, s.lda, s.ldv, 0, (int32_t)s.n, (int32_t)s.n, 1});
            big_idx.push_back(i);
            continue;
        }
        hd.push_back(CDesc{s.A, s.V, s.W, nullptr, s.lda, s.ldv, 0, (int32_t)s.n, (int32_t)s.n, 1});
        idx.push_back(i);
        lds = std::max(lds, lds_bytes(s.n, s.n));
    }
    return finish(ctx, hd, idx, lds, big, big_idx, info);
}

} // extern "C"
