// Batched SVD of small blocks: one workgroup per matrix, the whole one-sided Jacobi iteration in LDS.
//
// The block-Jacobi pipeline of svd_jacobi.hip pays two kernel launches per round; for the many <= 64-wide sector blocks
// of a small-chi or non-abelian (cfg4) tensor that is all latency (400 blocks of <= 40 x 40: 5.7 ms, 2.8x a scipy
// loop).  Here a matrix with min(m, n) <= 64 and max(m, n) <= 128 lives in LDS as the columns of W (the longer side as
// rows) next to the accumulated rotations V; a round rotates Np/2 disjoint column pairs at once -- eight lanes per pair:
// partial dot products over the rows, three __shfl_xor steps, the rotation applied to both columns of W and V -- and
// the rounds of a sweep follow the round-robin tournament.  No launch, no global traffic and one barrier per round.
// Reference semantics: scipy.linalg.svd(full_matrices=False) (NumpyBlockBackend::matrix_svd, numpy.cpp:1247-1297):
// S descending, U / Vh with orthonormal columns / rows also for rank-deficient blocks (null directions are completed
// by Gram-Schmidt against unit vectors).
#include "common.h"

#include <algorithm>
#include <vector>

namespace cyb {
int svd_small_batched(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* sweeps_out);
}

namespace {

constexpr int NT = 256;
constexpr int MAXN = 64, MAXM = 128;
constexpr int MAX_SWEEPS = 40;

struct SmallDesc {
    const double* A;
    double *U, *S, *Vh;
    int64_t lda, ldu, ldvh;
    int32_t m, n;
    int32_t mode, pad; // 0: SVD.  1: eigh of a symmetric block: S = eigenvalues (ascending), U = eigenvectors (may be NULL)
};

__device__ __forceinline__ double group8_sum(double v)
{
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

__device__ double block_sum(double v, double* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(NT) svd_small_kernel(const SmallDesc* __restrict__ descs, int32_t* __restrict__ sweeps_out)
{
    extern __shared__ double smem[];
    __shared__ double red[NT / 64];
    __shared__ double sig[MAXN];
    __shared__ int rank_of[MAXN];
    __shared__ double dots[MAXN];
    __shared__ int flag;
    const SmallDesc d = descs[blockIdx.x];
    const int tid = threadIdx.x;
    const bool tall = d.m >= d.n;
    const int M = tall ? d.m : d.n;          // rows of the work matrix
    const int N = tall ? d.n : d.m;          // its columns (the vectors that are orthogonalised)
    const int Np = (N + 1) & ~1;             // even number of columns (a zero column pads an odd count)
    const int ldw = M | 1, ldv = Np | 1;     // odd leading dimensions: the eight lanes of a pair hit distinct banks
    double* W = smem;                        // W[c * ldw + r]
    double* V = smem + (size_t)Np * ldw;     // V[c * ldv + r]
    // ---- load (transposed if the block is wide), V = identity
    for (int e = tid; e < Np * M; e += NT) {
        const int c = e / M, r = e - c * M;
        double v = 0.0;
        if (c < N) v = tall ? d.A[(int64_t)r * d.lda + c] : d.A[(int64_t)c * d.lda + r];
        W[c * ldw + r] = v;
    }
    for (int e = tid; e < Np * Np; e += NT) {
        const int c = e / Np, r = e - c * Np;
        V[c * ldv + r] = (c == r) ? 1.0 : 0.0;
    }
    __syncthreads();
    // columns whose norm falls below M eps ||A||_F are numerically null: they are not rotated any more (their direction
    // is rounding noise, which would never pass the relative test) and are completed by Gram-Schmidt at the end
    double fro2 = 0.0;
    for (int e = tid; e < Np * M; e += NT) {
        const int c = e / M, r = e - c * M;
        fro2 += W[c * ldw + r] * W[c * ldw + r];
    }
    fro2 = block_sum(fro2, red);
    // eigh: A + ||A||_F I is symmetric positive semi-definite, its right singular vectors are eigenvectors of A (any basis
    // of a degenerate eigenspace is one) and lambda = sigma - shift  (the shift of svd_jacobi.hip's eigh, in LDS)
    double shift = 0.0;
    if (d.mode == 1) {
        shift = sqrt(fro2);
        for (int c = tid; c < N; c += NT) W[c * ldw + c] += shift;
        __syncthreads();
        fro2 += shift * shift * N; // (bound for the null threshold)
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
                double* wp = W + p * ldw;
                double* wq = W + q * ldw;
                double app = 0.0, aqq = 0.0, apq = 0.0;
                for (int i = l8; i < M; i += 8) {
                    const double x = wp[i], y = wq[i];
                    app += x * x;
                    aqq += y * y;
                    apq += x * y;
                }
                app = group8_sum(app);
                aqq = group8_sum(aqq);
                apq = group8_sum(apq);
                const double den = sqrt(app) * sqrt(aqq); // (not sqrt(app * aqq): entries of 1e-80 .. 1e80 are in range)
                if (app > null2 && aqq > null2 && fabs(apq) > tol * den) {
                    off = fmax(off, fabs(apq) / den);
                    const double zeta = (aqq - app) / (2.0 * apq);
                    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    for (int i = l8; i < M; i += 8) {
                        const double x = wp[i], y = wq[i];
                        wp[i] = c * x - s * y;
                        wq[i] = s * x + c * y;
                    }
                    double* vp = V + p * ldv;
                    double* vq = V + q * ldv;
                    for (int i = l8; i < Np; i += 8) {
                        const double x = vp[i], y = vq[i];
                        vp[i] = c * x - s * y;
                        vq[i] = s * x + c * y;
                    }
                }
            }
            __syncthreads();
        }
        ++sweeps;
        // converged when a whole sweep needed no rotation
        const unsigned long long any = __ballot(off > 0.0);
        if (tid == 0) flag = 0;
        __syncthreads();
        if ((tid & 63) == 0 && any) flag = 1; // benign race: every writer stores 1
        __syncthreads();
        converged = (flag == 0);
        __syncthreads();
    }
    if (tid == 0) sweeps_out[blockIdx.x] = converged ? sweeps : -1;
    // ---- singular values, descending order
    for (int c = grp; c < Np; c += NT / 8) {
        double s2 = 0.0;
        for (int i = l8; i < M; i += 8) s2 += W[c * ldw + i] * W[c * ldw + i];
        s2 = group8_sum(s2);
        if (l8 == 0) sig[c] = (c < N) ? sqrt(s2) : -1.0; // the padding column sorts last
    }
    __syncthreads();
    if (tid < Np) {
        const double s = sig[tid];
        int rk = 0;
        for (int c = 0; c < Np; ++c) rk += (sig[c] > s) || (sig[c] == s && c < tid);
        rank_of[tid] = rk;
    }
    __syncthreads();
    if (d.mode == 1) {
        for (int c = tid; c < N; c += NT) d.S[N - 1 - rank_of[c]] = sig[c] - shift;
        if (d.U)
            for (int e = tid; e < N * N; e += NT) {
                const int j = e / N, c = e - j * N;
                d.U[(int64_t)j * d.ldu + (N - 1 - rank_of[c])] = V[c * ldv + j];
            }
        return;
    }
    const double thresh = sqrt(null2);
    // ---- normalise the left vectors; null directions are completed below
    for (int c = grp; c < N; c += NT / 8) {
        if (sig[c] > thresh) {
            const double inv = 1.0 / sig[c];
            for (int i = l8; i < M; i += 8) W[c * ldw + i] *= inv;
        }
    }
    __syncthreads();
    for (int c = 0; c < N; ++c) {
        if (sig[c] > thresh) continue; // (uniform: sig is in LDS)
        // complete column c: a unit vector orthogonalised against all finished columns (twice)
        bool done = false;
        for (int cand = 0; cand < M && !done; ++cand) {
            const int e = (cand + c) % M;
            // v = e_e - sum_k q_k[e] q_k over the finished columns k
            double v = 0.0;
            if (tid < M) {
                v = (tid == e) ? 1.0 : 0.0;
                for (int k = 0; k < N; ++k)
                    if (k != c && (sig[k] > thresh || k < c)) v -= W[k * ldw + e] * W[k * ldw + tid];
            }
            const double n2 = block_sum(v * v, red);
            // sum over all unit vectors of the squared residuals = M - (finished columns) >= 1, so some candidate has
            // at least 1 / M; with the second orthogonalisation pass that is far from any cancellation problem
            if (n2 * M > 0.5) { // (uniform)
                __syncthreads();
                if (tid < M) W[c * ldw + tid] = v;
                __syncthreads();
                for (int k = grp; k < N; k += NT / 8) {
                    double dt = 0.0;
                    if (k != c && (sig[k] > thresh || k < c))
                        for (int i = l8; i < M; i += 8) dt += W[k * ldw + i] * W[c * ldw + i];
                    dt = group8_sum(dt);
                    if (l8 == 0) dots[k] = dt;
                }
                __syncthreads();
                if (tid < M) {
                    for (int k = 0; k < N; ++k)
                        if (k != c && (sig[k] > thresh || k < c)) v -= dots[k] * W[k * ldw + tid];
                }
                const double n3 = block_sum(v * v, red);
                if (tid < M) W[c * ldw + tid] = v / sqrt(n3);
                __syncthreads();
                done = true;
            }
        }
    }
    __syncthreads();
    // ---- write U (m x k), S (k), Vh (k x n), k = N
    for (int c = tid; c < N; c += NT) d.S[rank_of[c]] = sig[c] > thresh ? sig[c] : fmax(sig[c], 0.0);
    if (tall) {
        for (int e = tid; e < N * M; e += NT) { // U[r, rank] = W[c][r]
            const int r = e / N, c = e - r * N;
            d.U[(int64_t)r * d.ldu + rank_of[c]] = W[c * ldw + r];
        }
        for (int e = tid; e < N * N; e += NT) { // Vh[rank, j] = V[c][j]
            const int c = e / N, j = e - c * N;
            d.Vh[(int64_t)rank_of[c] * d.ldvh + j] = V[c * ldv + j];
        }
    } else {
        for (int e = tid; e < N * N; e += NT) { // U[j, rank] = V[c][j]   (j < m = N)
            const int j = e / N, c = e - j * N;
            d.U[(int64_t)j * d.ldu + rank_of[c]] = V[c * ldv + j];
        }
        for (int e = tid; e < N * M; e += NT) { // Vh[rank, r] = W[c][r]  (r < n = M)
            const int c = e / M, r = e - c * M;
            d.Vh[(int64_t)rank_of[c] * d.ldvh + r] = W[c * ldw + r];
        }
    }
}

} // namespace

namespace cyb {

static int small_launch(cyb_ctx_t ctx, std::vector<SmallDesc>& hd, size_t lds, int32_t* sweeps_out);

bool svd_small_fits(int64_t m, int64_t n) { return std::min(m, n) <= MAXN && std::max(m, n) <= MAXM && std::min(m, n) >= 1; }

// sweeps_out[i] = sweeps used or -1 (host array of length n); synchronises.
int svd_small_batched(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* sweeps_out)
{
    if (n == 0) return CYB_OK;
    std::vector<SmallDesc> hd((size_t)n);
    size_t lds = 0;
    for (int64_t i = 0; i < n; ++i) {
        const cyb_svd_desc& s = descs[i];
        CYB_REQUIRE(svd_small_fits(s.m, s.n), "svd_small: block %lld (%lld x %lld) does not fit", (long long)i, (long long)s.m, (long long)s.n);
        CYB_REQUIRE(s.A && s.U && s.S && s.Vh, "svd block %lld: NULL pointer", (long long)i);
        CYB_REQUIRE(s.lda >= s.n && s.ldu >= std::min(s.m, s.n) && s.ldvh >= s.n,
                    "svd block %lld: leading dimension too small (lda=%lld ldu=%lld ldvh=%lld for %lld x %lld)", (long long)i,
                    (long long)s.lda, (long long)s.ldu, (long long)s.ldvh, (long long)s.m, (long long)s.n);
        hd[(size_t)i] = SmallDesc{s.A, s.U, s.S, s.Vh, s.lda, s.ldu, s.ldvh, (int32_t)s.m, (int32_t)s.n, 0, 0};
        const int M = (int)std::max(s.m, s.n), N = (int)std::min(s.m, s.n), Np = (N + 1) & ~1;
        lds = std::max(lds, sizeof(double) * ((size_t)Np * (M | 1) + (size_t)Np * (Np | 1)));
    }
    return small_launch(ctx, hd, lds, sweeps_out);
}

// eigh of symmetric blocks with n <= 64 (np.linalg.eigh semantics, numpy.cpp:658-680); V may be NULL (eigvalsh)
int eigh_small_batched(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* sweeps_out)
{
    if (n == 0) return CYB_OK;
    std::vector<SmallDesc> hd((size_t)n);
    size_t lds = 0;
    for (int64_t i = 0; i < n; ++i) {
        const cyb_eigh_desc& s = descs[i];
        CYB_REQUIRE(s.n >= 1 && s.n <= MAXN, "eigh_small: block %lld (n = %lld) does not fit", (long long)i, (long long)s.n);
        CYB_REQUIRE(s.A && s.W, "eigh block %lld: NULL pointer", (long long)i);
        CYB_REQUIRE(s.lda >= s.n && (!s.V || s.ldv >= s.n), "eigh block %lld: leading dimension too small", (long long)i);
        hd[(size_t)i] = SmallDesc{s.A, s.V, s.W, nullptr, s.lda, s.ldv, 0, (int32_t)s.n, (int32_t)s.n, 1, 0};
        const int N = (int)s.n, Np = (N + 1) & ~1;
        lds = std::max(lds, sizeof(double) * ((size_t)Np * (N | 1) + (size_t)Np * (Np | 1)));
    }
    return small_launch(ctx, hd, lds, sweeps_out);
}

static int small_launch(cyb_ctx_t ctx, std::vector<SmallDesc>& hd, size_t lds, int32_t* sweeps_out)
{
    const int64_t n = (int64_t)hd.size();
    static bool attr_set = false;
    if (!attr_set) {
        CYB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(svd_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(double) * ((size_t)MAXN * (MAXM | 1) + (size_t)MAXN * (MAXN | 1)))));
        attr_set = true;
    }
    void *d_descs = nullptr, *d_sw = nullptr;
    CYB_TRY(ctx->upload(hd.data(), sizeof(SmallDesc) * hd.size(), &d_descs));
    CYB_TRY(ctx->workspace(sizeof(int32_t) * (size_t)n, &d_sw, 3));
    hipLaunchKernelGGL(svd_small_kernel, dim3((unsigned)n), dim3(NT), lds, ctx->stream, static_cast<const SmallDesc*>(d_descs),
                       static_cast<int32_t*>(d_sw));
    CYB_HIP(hipGetLastError());
    std::vector<int32_t> sw((size_t)n);
    CYB_HIP(hipMemcpyAsync(sw.data(), d_sw, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    CYB_HIP(hipStreamSynchronize(ctx->stream));
    int st = CYB_OK;
    for (int64_t i = 0; i < n; ++i) {
        if (sweeps_out) sweeps_out[i] = sw[(size_t)i];
        if (sw[(size_t)i] < 0) st = CYB_ERR_NOCONV;
    }
    if (st == CYB_ERR_NOCONV) set_error("svd_small: a block did not converge in %d sweeps", MAX_SWEEPS);
    return st;
}

} // namespace cyb
