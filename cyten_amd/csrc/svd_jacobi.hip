// Batched thin SVD and symmetric eigendecomposition on the block-Jacobi engine.
//
//   cyb_svd_batched_f64   <-  NumpyBlockBackend::matrix_svd  (numpy.cpp:1247-1297), one call per
//                             sector block in AbelianBackend::svd (abelian.cpp:3517-3518)
//   cyb_eigh_batched_f64  <-  NumpyBlockBackend::eigh / eigvalsh (numpy.cpp:658-698)
//
// SVD of A (m x n):  nv = min(m,n) vectors of length len = max(m,n) are the rows of W0
//     m >= n :  W0 = A^T   ->  W = T W0 = S U^T,  Vh = T        (T accumulated in J)
//     m <  n :  W0 = A     ->  W = T W0 = S Vh ,  U  = T^T
// eigh of symmetric A:  W0 = A + c I  with  c = 2 |A|_F  is symmetric positive definite with
// condition number <= 3, so one-sided Jacobi converges in a few sweeps and the normalised rows of
// W ARE the eigenvectors (no accumulation needed);  lambda = sigma - c  (absolute accuracy
// ~ eps * |A|_F, the LAPACK dsyevd class of guarantee).
#include "blocked_qr.h"
#include "jacobi_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace cyb {
namespace {

#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS double* gp;
typedef const GLOBAL_AS double* gcp;

struct PrepDesc {
    const double* A;
    int64_t lda;
    int32_t m, n;
    double* W;
    double* J; // may be null
    int32_t nvp, lenp;
    int32_t transpose; // 1: W[j][i] = A[i][j];  0: W[i][j] = A[i][j]
    int32_t pad;
    const double* shift; // device scalar added on the diagonal (eigh), may be null
};

// W <- A or A^T (+ shift*I), 32x32 tiles through LDS so both sides are coalesced. grid.y = matrix
__global__ void __launch_bounds__(256) prep_kernel(const PrepDesc* __restrict__ descs)
{
    __shared__ double tile[32][33];
    const PrepDesc d = descs[blockIdx.y];
    gcp A = (gcp)d.A;
    gp W = (gp)d.W;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const int tiles_n = (d.n + 31) / 32, tiles_m = (d.m + 31) / 32;
    const double shift = d.shift ? *(gcp)d.shift : 0.0;
    for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
        const int ti = t / tiles_n, tj = t % tiles_n;
        const int i0 = ti * 32, j0 = tj * 32;
        if (d.transpose) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + ty + 8 * r, j = j0 + tx;
                tile[ty + 8 * r][tx] = (i < d.m && j < d.n) ? A[(int64_t)i * d.lda + j] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + ty + 8 * r, i = i0 + tx;
                if (j < d.n && i < d.m) W[(int64_t)j * d.lenp + i] = tile[tx][ty + 8 * r];
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + ty + 8 * r, j = j0 + tx;
                if (i < d.m && j < d.n) {
                    double v = A[(int64_t)i * d.lda + j];
                    if (i == j) v += shift;
                    W[(int64_t)i * d.lenp + j] = v;
                }
            }
        }
    }
    if (d.J) {
        gp J = (gp)d.J;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < d.nvp; i += gridDim.x * 256) J[(int64_t)i * d.nvp + i] = 1.0;
    }
}

// out[b] = 2 * ||A_b||_F  (one workgroup per matrix)
struct NormDesc {
    const double* A;
    int64_t lda;
    int32_t m, n;
    double* out;
};
__global__ void __launch_bounds__(256) fro_shift_kernel(const NormDesc* __restrict__ descs)
{
    __shared__ double red[4];
    const NormDesc d = descs[blockIdx.x];
    gcp A = (gcp)d.A;
    double s = 0.0;
    const int64_t tot = (int64_t)d.m * d.n;
    for (int64_t e = threadIdx.x; e < tot; e += 256) {
        const double v = A[(e / d.n) * d.lda + (e % d.n)];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double f = sqrt(red[0] + red[1] + red[2] + red[3]);
        *(gp)d.out = (f > 0.0) ? 2.0 * f : 1.0;
    }
}

struct PostDesc {
    const double* W;
    const double* J;
    int32_t nvp, lenp, nv, len;
    int32_t transposed; // SVD: 1 if m >= n
    int32_t mode;       // 0 = SVD (descending), 1 = eigh (ascending, lambda = sigma - shift)
    double* sig;        // nv  : row norms (workspace)
    int32_t* rank;      // nv  : position of vector j in the sorted output (workspace)
    double* S;          // nv  : sorted singular values / eigenvalues (output)
    double* U;          // SVD: m x k ; eigh: V (n x n) or null
    int64_t ldu;
    double* Vh;         // SVD: k x n ; eigh: unused
    int64_t ldvh;
    const double* shift;
    double* scratch;    // lenp doubles (null-space completion)
    int32_t* n_null;    // device counter of numerically zero singular values (SVD only)
    double null_scale;  // factor applied to the W rows of zero singular values: 0 (completed later) or 1 (already unit)
    int32_t cplx = 0;   // rows 2a, 2a+1 are the interleaved embedding of complex row a: the pair is ranked by the even row's value
    int32_t pad_ = 0;
};

// sig[j] = || W[j,:] ||, one wave per row.  grid.y = matrix
__global__ void __launch_bounds__(256) row_norm_kernel(const PostDesc* __restrict__ descs)
{
    const PostDesc d = descs[blockIdx.y];
    gcp W = (gcp)d.W;
    const int lane = threadIdx.x & 63;
    for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < d.nv; j += gridDim.x * 4) {
        // two-pass scaled norm is not needed: entries are O(|A|); plain sum of squares in fp64
        double s = 0.0;
        for (int c = lane; c < d.len; c += 64) {
            const double v = W[(int64_t)j * d.lenp + c];
            s += v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) ((gp)d.sig)[j] = sqrt(s);
    }
}

// rank[j] = position of sig[j] in sorted order; S[rank[j]] = value.  grid.y = matrix
__global__ void __launch_bounds__(256) rank_kernel(const PostDesc* __restrict__ descs)
{
    constexpr int CH = 2048; // keys staged per pass (a thread comparing against nv values in global memory took 0.13 ms at nv = 824)
    __shared__ double keys[CH];
    const PostDesc d = descs[blockIdx.y];
    gcp sig = (gcp)d.sig;
    const double shift = d.shift ? *(gcp)d.shift : 0.0;
    const int msk = d.cplx ? ~1 : ~0;
    const int per = (d.nv + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256); // values per thread
    for (int it = 0; it < per; ++it) {
        const int j = (it * (int)gridDim.x + (int)blockIdx.x) * 256 + (int)threadIdx.x;
        const bool mine = j < d.nv;
        const double sj = mine ? sig[j & msk] : 0.0;
        int r = 0;
        for (int k0 = 0; k0 < d.nv; k0 += CH) {
            const int kn = min(CH, d.nv - k0);
            __syncthreads();
            for (int k = threadIdx.x; k < kn; k += 256) keys[k] = sig[(k0 + k) & msk];
            __syncthreads();
            if (mine) {
                if (d.mode == 0) {
                    for (int k = 0; k < kn; ++k) {
                        const double sk = keys[k];
                        r += (sk > sj || (sk == sj && k0 + k < j)) ? 1 : 0;
                    }
                } else {
                    for (int k = 0; k < kn; ++k) {
                        const double sk = keys[k];
                        r += (sk < sj || (sk == sj && k0 + k < j)) ? 1 : 0;
                    }
                }
            }
        }
        if (mine) {
            d.rank[j] = r;
            d.S[r] = (d.mode == 0) ? sj : sj - shift;
        }
    }
}

// Write the factors in sorted order.  grid.y = matrix; grid.x strides over 32x32 tiles.
//   "row side":    out_r[rank j][c] = src[j][c] * scale      (coalesced both ways)
//   "column side": out_c[i][rank j] = src[j][i] * scale      (tile transpose through LDS)
__global__ void __launch_bounds__(256) write_factors_kernel(const PostDesc* __restrict__ descs)
{
    __shared__ double tile[32][33];
    __shared__ int rk[32];
    __shared__ double sc[32];
    const PostDesc d = descs[blockIdx.y];
    gcp W = (gcp)d.W;
    gcp J = (gcp)d.J;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // One-sided Jacobi orthogonalises rows RELATIVE to their norms, so even rows at rounding-noise
    // level (numerically zero singular values) normalise to an orthonormal set.  Only rows that
    // are exactly zero (or so small that 1/sigma overflows) have no direction: those get completed.
    double smax = 0.0;
    if (d.mode == 0) smax = ((gcp)d.S)[0];
    const double thresh = fmax(smax * 1e-280, 1e-300);

    // which source feeds the column side / row side
    //  SVD, m >= n : U[i][r] = W[j][i]/s (column side, from W, len = m), Vh[r][c] = J[j][c] (row side, nv cols)
    //  SVD, m <  n : U[i][r] = J[j][i]   (column side, from J, nv rows), Vh[r][c] = W[j][c]/s (row side, len cols)
    //  eigh        : V[i][r] = W[j][i]/s (column side)
    const bool col_from_W = (d.mode == 1) || d.transposed;
    gcp csrc = col_from_W ? W : J;
    const int csrc_ld = col_from_W ? d.lenp : d.nvp;
    const int c_rows = col_from_W ? d.len : d.nv; // number of i
    gcp rsrc = col_from_W ? J : W;
    const int rsrc_ld = col_from_W ? d.nvp : d.lenp;
    const int r_cols = col_from_W ? d.nv : d.len;

    const int tj = (d.nv + 31) / 32;
    // ---- column side
    if (d.U) {
        const int ti_n = (c_rows + 31) / 32;
        for (int t = blockIdx.x; t < tj * ti_n; t += gridDim.x) {
            const int j0 = (t / ti_n) * 32, i0 = (t % ti_n) * 32;
            __syncthreads();
            if (threadIdx.x < 32) {
                const int j = j0 + threadIdx.x;
                double s = 1.0;
                int r = 0;
                if (j < d.nv) {
                    r = d.rank[j];
                    if (col_from_W) {
                        const double sj = ((gcp)d.sig)[j];
                        s = (d.mode == 0 && sj <= thresh) ? d.null_scale : 1.0 / sj;
                    }
                }
                rk[threadIdx.x] = r;
                sc[threadIdx.x] = s;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + ty + 8 * q, i = i0 + tx;
                tile[ty + 8 * q][tx] = (j < d.nv && i < c_rows) ? csrc[(int64_t)j * csrc_ld + i] * sc[ty + 8 * q] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + ty + 8 * q, j = j0 + tx;
                if (i < c_rows && j < d.nv) ((gp)d.U)[(int64_t)i * d.ldu + rk[tx]] = tile[tx][ty + 8 * q];
            }
        }
    }
    // ---- row side (SVD only)
    if (d.mode == 0 && d.Vh) {
        for (int j = blockIdx.x; j < d.nv; j += gridDim.x) {
            const int r = d.rank[j];
            double s = 1.0;
            if (!col_from_W) {
                const double sj = ((gcp)d.sig)[j];
                s = (sj <= thresh) ? d.null_scale : 1.0 / sj;
            }
            for (int c = threadIdx.x; c < r_cols; c += 256)
                ((gp)d.Vh)[(int64_t)r * d.ldvh + c] = rsrc[(int64_t)j * rsrc_ld + c] * s;
        }
    }
    if (d.mode == 0 && blockIdx.x == 0) { // (the whole workgroup counts: one thread walking nv values took 0.18 ms at nv = 824)
        __syncthreads();
        int cnt = 0;
        for (int j = threadIdx.x; j < d.nv; j += 256) cnt += (((gcp)d.sig)[j] <= thresh) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if ((threadIdx.x & 63) == 0) rk[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) *d.n_null = rk[0] + rk[1] + rk[2] + rk[3];
    }
}

// Orthonormal completion for exactly-zero singular values (e.g. zero or exactly rank-deficient
// blocks): the accumulated factor (J side) is orthogonal by construction, the normalised W side
// has no direction for sigma = 0.  For every such vector take the unit vector e_t with the largest
// residual 1 - sum_q F[t,q]^2 w.r.t. the vectors defined so far (a pivoted choice: the residuals
// sum to len - #defined >= 1, so the best one is never tiny), orthogonalise it with two
// Gram-Schmidt passes and normalise.  One workgroup per matrix; rare path.
__global__ void __launch_bounds__(256) complete_null_kernel(const PostDesc* __restrict__ descs)
{
    __shared__ double red[4];
    __shared__ int redi[4];
    __shared__ int s_pick;
    const PostDesc d = descs[blockIdx.x];
    if (d.mode != 0 || *d.n_null == 0) return;
    const int tid = threadIdx.x;
    // the W-side factor: nv vectors of length len; vector at sorted position r:
    //   m >= n : column r of U   (element i at U[i*ldu + r])
    //   m <  n : row r of Vh     (element c at Vh[r*ldvh + c])
    gp F = (gp)(d.transposed ? d.U : d.Vh);
    if (!F) return;
    const int64_t es = d.transposed ? d.ldu : 1;   // stride between elements of one vector
    const int64_t vs = d.transposed ? 1 : d.ldvh;  // stride between vectors
    gp v = (gp)d.scratch;        // len doubles: the candidate
    gp resid = v + d.lenp;       // len doubles: residual of every unit vector
    const int n_null = *d.n_null;
    const int first_null = d.nv - n_null; // sorted descending: the zeros are the trailing positions
    for (int i = tid; i < d.len; i += 256) {
        double s = 1.0;
        for (int q = 0; q < first_null; ++q) {
            const double f = F[(int64_t)i * es + q * vs];
            s -= f * f;
        }
        resid[i] = s;
    }
    __syncthreads();
    for (int r = first_null; r < d.nv; ++r) {
        // argmax of the residuals
        double best = -1.0;
        int bi = 0;
        for (int i = tid; i < d.len; i += 256)
            if (resid[i] > best) {
                best = resid[i];
                bi = i;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) {
                best = ob;
                bi = oi;
            }
        }
        if ((tid & 63) == 0) {
            red[tid >> 6] = best;
            redi[tid >> 6] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            int p = 0;
            for (int w = 1; w < 4; ++w)
                if (red[w] > red[p] || (red[w] == red[p] && redi[w] < redi[p])) p = w;
            s_pick = redi[p];
        }
        __syncthreads();
        const int cand = s_pick;
        for (int i = tid; i < d.len; i += 256) v[i] = (i == cand) ? 1.0 : 0.0;
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass) {
            for (int q = 0; q < r; ++q) {
                double s = 0.0;
                for (int i = tid; i < d.len; i += 256) s += F[(int64_t)i * es + q * vs] * v[i];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                __syncthreads();
                if ((tid & 63) == 0) red[tid >> 6] = s;
                __syncthreads();
                const double dot = red[0] + red[1] + red[2] + red[3];
                for (int i = tid; i < d.len; i += 256) v[i] -= dot * F[(int64_t)i * es + q * vs];
                __syncthreads();
            }
        }
        double s = 0.0;
        for (int i = tid; i < d.len; i += 256) s += v[i] * v[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        const double nrm2 = red[0] + red[1] + red[2] + red[3];
        const double inv = (nrm2 > 0.0) ? 1.0 / sqrt(nrm2) : 0.0;
        for (int i = tid; i < d.len; i += 256) {
            const double f = v[i] * inv;
            F[(int64_t)i * es + r * vs] = f;
            resid[i] = (i == cand) ? -1.0 : resid[i] - f * f;
        }
        __syncthreads();
    }
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct Layout {
    std::vector<JMat> mats;
    std::vector<PrepDesc> prep;
    std::vector<PostDesc> post;
    size_t bytes = 0;
};

} // namespace

// Row blocks the iteration pairs up: enough 16-row blocks for the nv vectors in use, an even number, NOT the padded count
// nvp / JB (nvp is a multiple of 64 because the rows of J are whole 64-column chunks): 721 vectors are 46 blocks = 45 rounds
// per sweep, not 48 = 47.  The rows between nb * JB and nvp are zero in W and unit rows in J and never move.
static inline int row_blocks(int nv, int nvp)
{
    const int nb = 2 * ((nv + 2 * cyb::JB - 1) / (2 * cyb::JB));
    return std::min(std::max(nb, 4), nvp / cyb::JB);
}

// shared driver: mode 0 = SVD, 1 = eigh
static int run_jacobi(cyb_ctx_t ctx, int mode, int64_t nmat, const cyb_svd_desc* sd, const cyb_eigh_desc* ed,
                      int32_t* info, bool cplx = false)
{   // cplx (mode 1 only): the blocks are interleaved embeddings of complex Hermitian blocks
    if (nmat == 0) return CYB_OK;
    hipStream_t st = ctx->stream;
    // ---- workspace layout
    std::vector<JMat> mats((size_t)nmat);
    std::vector<PrepDesc> prep((size_t)nmat);
    std::vector<PostDesc> post((size_t)nmat);
    std::vector<NormDesc> nd;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    struct Offs {
        size_t W, J, sig, rank, shift, scratch, nnull;
    };
    std::vector<Offs> offs((size_t)nmat);
    for (int64_t b = 0; b < nmat; ++b) {
        int m, n;
        if (mode == 0) {
            CYB_REQUIRE(sd[b].m >= 0 && sd[b].n >= 0 && sd[b].m < (1 << 30) && sd[b].n < (1 << 30),
                        "svd block %lld: bad shape", (long long)b);
            m = (int)sd[b].m;
            n = (int)sd[b].n;
            CYB_REQUIRE(m == 0 || n == 0 || (sd[b].A && sd[b].S), "svd block %lld: NULL pointer", (long long)b);
            CYB_REQUIRE(sd[b].lda >= n, "svd block %lld: lda < n", (long long)b);
        } else {
            CYB_REQUIRE(ed[b].n >= 0 && ed[b].n < (1 << 30), "eigh block %lld: bad shape", (long long)b);
            m = n = (int)ed[b].n;
            CYB_REQUIRE(n == 0 || (ed[b].A && ed[b].W), "eigh block %lld: NULL pointer", (long long)b);
            CYB_REQUIRE(ed[b].lda >= n, "eigh block %lld: lda < n", (long long)b);
        }
        const int nv = std::min(m, n), len = std::max(m, n);
        const int nvp = std::max(round_up(nv, 64), 64), lenp = std::max(round_up(len, 64), 64);
        const bool need_J = (mode == 0);
        Offs& o = offs[(size_t)b];
        o.W = take(sizeof(double) * (size_t)nvp * lenp);
        o.J = need_J ? take(sizeof(double) * (size_t)nvp * nvp) : 0;
        o.sig = take(sizeof(double) * (size_t)nvp);
        o.rank = take(sizeof(int32_t) * (size_t)nvp);
        o.shift = take(sizeof(double));
        o.scratch = take(sizeof(double) * (size_t)lenp * 2);
        o.nnull = take(sizeof(int32_t));
        JMat& jm = mats[(size_t)b];
        jm.nvp = nvp;
        jm.lenp = lenp;
        jm.nb = row_blocks(nv, nvp);
        jm.nv = nv;
        jm.len = len;
        jm.pad = 0;
        jm.tol = 2.220446049250313e-16 * std::max(16.0, 4.0 * std::sqrt((double)len));
        jm.thr2 = 0.0;
    }
    void* ws = nullptr;
    CYB_TRY(ctx->workspace(off, &ws));
    char* base = static_cast<char*>(ws);
    CYB_HIP(hipMemsetAsync(ws, 0, off, st));
    for (int64_t b = 0; b < nmat; ++b) {
        const Offs& o = offs[(size_t)b];
        JMat& jm = mats[(size_t)b];
        jm.W = reinterpret_cast<double*>(base + o.W);
        jm.J = (mode == 0) ? reinterpret_cast<double*>(base + o.J) : nullptr;
        PrepDesc& p = prep[(size_t)b];
        PostDesc& q = post[(size_t)b];
        if (mode == 0) {
            p.A = sd[b].A;
            p.lda = sd[b].lda;
            p.m = (int)sd[b].m;
            p.n = (int)sd[b].n;
            p.transpose = (sd[b].m >= sd[b].n) ? 1 : 0;
            p.shift = nullptr;
            q.U = sd[b].U;
            q.ldu = sd[b].ldu;
            q.Vh = sd[b].Vh;
            q.ldvh = sd[b].ldvh;
            q.S = sd[b].S;
            q.shift = nullptr;
        } else {
            p.A = ed[b].A;
            p.lda = ed[b].lda;
            p.m = p.n = (int)ed[b].n;
            p.transpose = 0;
            p.shift = reinterpret_cast<double*>(base + o.shift);
            q.U = ed[b].V;
            q.ldu = ed[b].ldv;
            q.Vh = nullptr;
            q.ldvh = 0;
            q.S = ed[b].W;
            q.shift = p.shift;
            nd.push_back(NormDesc{ed[b].A, ed[b].lda, (int)ed[b].n, (int)ed[b].n, reinterpret_cast<double*>(base + o.shift)});
        }
        p.W = jm.W;
        p.J = jm.J;
        p.nvp = jm.nvp;
        p.lenp = jm.lenp;
        p.pad = 0;
        q.W = jm.W;
        q.J = jm.J;
        q.nvp = jm.nvp;
        q.lenp = jm.lenp;
        q.nv = jm.nv;
        q.len = jm.len;
        q.transposed = p.transpose;
        q.mode = mode;
        q.sig = reinterpret_cast<double*>(base + o.sig);
        q.rank = reinterpret_cast<int32_t*>(base + o.rank);
        q.scratch = reinterpret_cast<double*>(base + o.scratch);
        q.n_null = reinterpret_cast<int32_t*>(base + o.nnull);
        q.null_scale = 0.0;
        q.cplx = cplx ? 1 : 0;
    }
    // ---- prepare W (and J = I)
    void* d_prep = nullptr;
    CYB_TRY(ctx->upload(prep.data(), sizeof(PrepDesc) * prep.size(), &d_prep));
    if (mode == 1) {
        void* d_nd = nullptr;
        CYB_TRY(ctx->upload(nd.data(), sizeof(NormDesc) * nd.size(), &d_nd));
        hipLaunchKernelGGL(fro_shift_kernel, dim3((unsigned)nmat), dim3(256), 0, st, static_cast<const NormDesc*>(d_nd));
    }
    hipLaunchKernelGGL(prep_kernel, dim3(64, (unsigned)nmat), dim3(256), 0, st, static_cast<const PrepDesc*>(d_prep));
    CYB_HIP(hipGetLastError());
    // ---- orthogonalise
    std::vector<int32_t> sweeps;
    const int jst = jacobi_orthogonalise(ctx, mats, 40, sweeps, cplx);
    if (info)
        for (int64_t b = 0; b < nmat; ++b) info[b] = sweeps[(size_t)b];
    if (jst != CYB_OK && jst != CYB_ERR_NOCONV) return jst;
    // ---- read off the factors
    void* d_post = nullptr;
    CYB_TRY(ctx->upload(post.data(), sizeof(PostDesc) * post.size(), &d_post));
    const PostDesc* dp = static_cast<const PostDesc*>(d_post);
    hipLaunchKernelGGL(row_norm_kernel, dim3(helper_grid_x((size_t)nmat), (unsigned)nmat), dim3(256), 0, st, dp);
    hipLaunchKernelGGL(rank_kernel, dim3(8, (unsigned)nmat), dim3(256), 0, st, dp);
    hipLaunchKernelGGL(write_factors_kernel, dim3(helper_grid_x((size_t)nmat), (unsigned)nmat), dim3(256), 0, st, dp);
    if (mode == 0) hipLaunchKernelGGL(complete_null_kernel, dim3((unsigned)nmat), dim3(256), 0, st, dp);
    CYB_HIP(hipGetLastError());
    if (info) CYB_HIP(hipStreamSynchronize(st));
    return jst;
}


// ================================================================================================
// SVD pipeline v2:  QR preconditioning -> block Jacobi with deflation -> null completion -> assembly
//
//   tall (m >= n):  A   = Q1 R,  Jacobi on the rows of R:  T R = S X  =>  U = Q1 T^T,  Vh = X
//   wide (m <  n):  A^T = Q1 R,                            T R = S X  =>  U = X^T,     Vh = (Q1 T^T)^T
//
// Why (measured): (i) a tall block shrinks to min(m,n)^2 before the
// iteration; (ii) for rank-deficient blocks -- every block of a theta = A.B -- the trailing rows
// of R are at rounding-noise level from the start, so the threshold deflation of the round kernel
// removes them in the first sweep and the iteration runs on the numerical rank only (8 sweeps on
// half the rows instead of 27 on all of them at chi = 4096); (iii) the orthonormal completion of
// the deflated directions comes from the trailing columns of a full Householder Q (exactly
// orthonormal) instead of from iterating on noise.
namespace {

struct RowsDesc {
    double* W;          // kp x kp
    const double* sig;
    const int32_t* idx; // row indices (device)
    double* buf;        // col-major k x cnt (ld = k)
    int32_t kp, k, cnt, pad;
    const int32_t* col; // scatter only: column of buf that goes to row idx[c] (null: column c)
};
// buf[:, c] = W[idx[c], 0:k] / sig[idx[c]]    (grid.y = matrix)
__global__ void __launch_bounds__(256) gather_rows_kernel(const RowsDesc* __restrict__ descs)
{
    const RowsDesc d = descs[blockIdx.y];
    gcp W = (gcp)d.W;
    gp buf = (gp)d.buf;
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < d.cnt; c += gridDim.x * 4) {
        const int j = d.idx[c];
        const double inv = 1.0 / ((gcp)d.sig)[j];
        for (int i = lane; i < d.k; i += 64) buf[(int64_t)c * d.k + i] = W[(int64_t)j * d.kp + i] * inv;
    }
}
// W[idx[c], 0:k] = buf[:, c]
__global__ void __launch_bounds__(256) scatter_rows_kernel(const RowsDesc* __restrict__ descs)
{
    const RowsDesc d = descs[blockIdx.y];
    gp W = (gp)d.W;
    gcp buf = (gcp)d.buf;
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < d.cnt; c += gridDim.x * 4) {
        const int j = d.idx[c];
        const int cc = d.col ? d.col[c] : c;
        for (int i = lane; i < d.k; i += 64) W[(int64_t)j * d.kp + i] = buf[(int64_t)cc * d.k + i];
    }
}

struct JcqDesc {
    const double* J;
    const int32_t* rank;
    double* Cq; // col-major L x k
    int64_t L;
    int32_t kp, k;
};
// Cq[:, rank[j]] = [ J[j, 0:k] ; 0 ]
__global__ void __launch_bounds__(256) j_to_cq_kernel(const JcqDesc* __restrict__ descs)
{
    const JcqDesc d = descs[blockIdx.y];
    gcp J = (gcp)d.J;
    gp Cq = (gp)d.Cq;
    const int lane = threadIdx.x & 63;
    for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < d.k; j += gridDim.x * 4) {
        const int r = d.rank[j];
        for (int i = lane; i < d.k; i += 64) Cq[(int64_t)r * d.L + i] = J[(int64_t)j * d.kp + i];
    }
}


struct RowMoveDesc {
    const double* src;
    double* dst;
    const int32_t* idx;
    int64_t lds, ldd;
    int32_t cnt, ncols;
    int32_t mode, pad; // 0: dst[g] = src[idx[g]]   1: dst[idx[g]] = src[g]   2: dst[idx[g]] = 0
};
__global__ void __launch_bounds__(256) row_move_kernel(const RowMoveDesc* __restrict__ descs)
{
    const RowMoveDesc d = descs[blockIdx.y];
    gcp src = (gcp)d.src;
    gp dst = (gp)d.dst;
    const int lane = threadIdx.x & 63;
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g < d.cnt; g += gridDim.x * 4) {
        const int j = d.idx[g];
        if (d.mode == 0) {
            for (int c = lane; c < d.ncols; c += 64) dst[(int64_t)g * d.ldd + c] = src[(int64_t)j * d.lds + c];
        } else if (d.mode == 1) {
            for (int c = lane; c < d.ncols; c += 64) dst[(int64_t)j * d.ldd + c] = src[(int64_t)g * d.lds + c];
        } else {
            for (int c = lane; c < d.ncols; c += 64) dst[(int64_t)j * d.ldd + c] = 0.0;
        }
    }
}
// J[idx[g]][idx[h]] = Jc[g][h]   (g, h < cnt)
__global__ void __launch_bounds__(256) j_scatter_kernel(const RowMoveDesc* __restrict__ descs)
{
    const RowMoveDesc d = descs[blockIdx.y];
    gcp src = (gcp)d.src;
    gp dst = (gp)d.dst;
    const int lane = threadIdx.x & 63;
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g < d.cnt; g += gridDim.x * 4) {
        const int j = d.idx[g];
        for (int h = lane; h < d.cnt; h += 64) dst[(int64_t)j * d.ldd + d.idx[h]] = src[(int64_t)g * d.lds + h];
    }
}

// ---- embedded complex rows: rows 2a = [x, -y] and 2a+1 = [y, x] (interleaved) of the iteration matrix are made EXACT
//      partners before the first sweep (both copies averaged).  The QR steps leave them partners up to eps ||A|| -- for the
//      small rows of a graded spectrum that is a large RELATIVE defect (1e-4 at sigma = 1e-12 sigma_max), the structured
//      pivot solves never touch the coupling inside a pair, and the iteration would stall on it; the change itself is a
//      backward error of eps ||A||.  The sweeps then keep the partners to relative rounding (their updates use the same
//      coefficients on partner data), like every other row relation of a one-sided Jacobi iteration.
struct PairDesc {
    double* W;
    int64_t ld;
    int32_t nv, len; // rows and columns in use (both even)
};
__global__ void __launch_bounds__(256) embed_pairs_kernel(const PairDesc* __restrict__ descs)
{
    const PairDesc d = descs[blockIdx.y];
    gp W = (gp)d.W;
    const int lane = threadIdx.x & 63;
    for (int a = blockIdx.x * 4 + (threadIdx.x >> 6); a < d.nv / 2; a += gridDim.x * 4) {
        gp e = W + (int64_t)(2 * a) * d.ld, o = e + d.ld;
        for (int j = lane; j < d.len / 2; j += 64) {
            const double x = 0.5 * (e[2 * j] + o[2 * j + 1]), y = 0.5 * (o[2 * j] - e[2 * j + 1]);
            e[2 * j] = x;
            e[2 * j + 1] = -y;
            o[2 * j] = y;
            o[2 * j + 1] = x;
        }
    }
}

// ---- second (LQ) preconditioning step: kernels that move between the iteration on R2 and the (W, J) pair the
//      read-off expects (run_svd_qr, step 2c)
struct LqDesc {
    double* Wq;   // rp x rp row-major: R2 before the iteration, S Z^T after it
    double* Jc;   // rp x rp row-major: accumulated rotations J' in, Z^T (normalised rows of Wq) out
    double* Wc;   // r0 x kp row-major: sigma_i * (column i of Cq2) out
    double* Cq2;  // k x k col-major (ld = k): [ J'^T ; 0 | unit vectors r0 .. k-1 ] -> Q2 applied in place
    double* sig2; // r0 row norms of Wq
    double* bad;  // set to 1 when a row of Wq is numerically null (no left vector to read off)
    double thr2;  // rows with sigma^2 <= thr2 are numerically null: their Wc row is zeroed (completed later)
    int32_t rp, kp, k, r0;
    int32_t cplx, pad_; // embedded complex rows: a pair is null or not as a whole
};
// Cq2[:, i] = [ J'[i, 0:r0] ; 0 ] (i < r0),  Cq2[:, t] = e_t (r0 <= t < k)
__global__ void __launch_bounds__(256) lq_pack_kernel(const LqDesc* __restrict__ descs)
{
    const LqDesc d = descs[blockIdx.y];
    gcp Jc = (gcp)d.Jc;
    gp C = (gp)d.Cq2;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < d.k; i += gridDim.x * 4) {
        for (int c = lane; c < d.k; c += 64) {
            double v;
            if (i < d.r0) v = c < d.r0 ? Jc[(int64_t)i * d.rp + c] : 0.0;
            else v = c == i ? 1.0 : 0.0;
            C[(int64_t)i * d.k + c] = v;
        }
    }
}
// sig2[i] = || Wq[i, :] ||,  Jc[i, 0:r0] = Wq[i, 0:r0] / sig2[i]
__global__ void __launch_bounds__(256) lq_rows_kernel(const LqDesc* __restrict__ descs)
{
    const LqDesc d = descs[blockIdx.y];
    gcp Wq = (gcp)d.Wq;
    gp Jc = (gp)d.Jc;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < d.r0; i += gridDim.x * 4) {
        double s = 0.0;
        for (int c = lane; c < d.r0; c += 64) {
            const double v = Wq[(int64_t)i * d.rp + c];
            s += v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const double sg = sqrt(s);
        if (lane == 0) {
            ((gp)d.sig2)[i] = sg;
            if (!(sg > 0.0)) *(gp)d.bad = 1.0;
        }
        const double inv = sg > 0.0 ? 1.0 / sg : 0.0;
        for (int c = lane; c < d.r0; c += 64) Jc[(int64_t)i * d.rp + c] = Wq[(int64_t)i * d.rp + c] * inv;
    }
}
// A row at the rounding level of the matrix (sigma_i^2 <= thr2) may be a legitimate small singular value -- its direction
// is then as orthogonal to the others as any (the iteration orthogonalises relative to the norms) -- or pure rounding noise
// of a rank deficiency the row norms of R did not show (zero columns of A: R has zero columns but no small row), and
// normalised noise is orthogonal to nothing (|V V^T - 1| = 0.8 on a 462 x 600 block with 140 zero rows, scripts/svd_fuzz.py
// seed 53).  Measured directly: the Gram row of every such direction; above 1e-12 the list takes the plain iteration.
__global__ void __launch_bounds__(256) lq_check_kernel(const LqDesc* __restrict__ descs)
{
    const LqDesc d = descs[blockIdx.y];
    gcp Jc = (gcp)d.Jc;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    // every wave finds the flagged rows (one ballot per 64 rows) and takes its share of the Gram row of each
    for (int i0 = 0; i0 < d.r0; i0 += 64) {
        bool fl = false;
        if (i0 + lane < d.r0) {
            const double sg = ((gcp)d.sig2)[i0 + lane];
            fl = !(sg * sg > d.thr2);
        }
        unsigned long long msk = __ballot(fl);
        while (msk) {
            const int i = i0 + __ffsll((long long)msk) - 1;
            msk &= msk - 1;
            double worst = 0.0;
            for (int j = wid; j < d.r0; j += nw) {
                if (j == i) continue;
                double t = 0.0;
                for (int c = lane; c < d.r0; c += 64) t = fma(Jc[(int64_t)i * d.rp + c], Jc[(int64_t)j * d.rp + c], t);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
                worst = fmax(worst, fabs(t));
            }
            if (lane == 0 && !(worst <= 1e-12)) *(gp)d.bad = 1.0;
        }
    }
}
// Wc[i, 0:k] = sigma_i * Cq2[0:k, i]   (zero for numerically null rows)
__global__ void __launch_bounds__(256) lq_unpack_kernel(const LqDesc* __restrict__ descs)
{
    const LqDesc d = descs[blockIdx.y];
    gcp C = (gcp)d.Cq2;
    gp Wc = (gp)d.Wc;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < d.r0; i += gridDim.x * 4) {
        const double sg = ((gcp)d.sig2)[i];
        const double sm = d.cplx ? fmax(sg, ((gcp)d.sig2)[i ^ 1]) : sg;
        const double f = sm * sm > d.thr2 ? sg : 0.0;
        for (int c = lane; c < d.kp; c += 64) Wc[(int64_t)i * d.kp + c] = c < d.k ? f * C[(int64_t)i * d.k + c] : 0.0;
    }
}

static int launch_row_moves(cyb_ctx_t ctx, const std::vector<RowMoveDesc>& v, bool jscatter = false)
{
    if (v.empty()) return CYB_OK;
    void* d = nullptr;
    CYB_TRY(ctx->upload(v.data(), sizeof(RowMoveDesc) * v.size(), &d));
    if (jscatter)
        hipLaunchKernelGGL(j_scatter_kernel, dim3(64, (unsigned)v.size()), dim3(256), 0, ctx->stream, static_cast<const RowMoveDesc*>(d));
    else
        hipLaunchKernelGGL(row_move_kernel, dim3(helper_grid_x(v.size()), (unsigned)v.size()), dim3(256), 0, ctx->stream, static_cast<const RowMoveDesc*>(d));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

} // namespace

static int run_svd_qr(cyb_ctx_t ctx, int64_t nmat, const cyb_svd_desc* sd, int32_t* info, int flags, int32_t* rank_out)
{
    if (nmat == 0) return CYB_OK;
    // CYB_SVD_EMBEDDED_COMPLEX: every block is the interleaved real embedding M(A) of a complex block (entry a + ib ->
    // [[a, -b], [b, a]]).  The QR steps preserve the structure by themselves (uniqueness); the iteration does through its
    // structure-preserving pivot solve (jacobi_engine.hip); here rows 2a, 2a + 1 are deflated, ranked and completed as pairs.
    const bool cplx = (flags & CYB_SVD_EMBEDDED_COMPLEX) != 0;
    hipStream_t st = ctx->stream;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    struct Lay {
        int m, n, k, L, kp;
        bool tall;
        size_t Ac, aux, W, J, sig, rank, thr, nnull, Cq, Fc, aux2, Cn, idx, Wc, Jc;
        size_t bad, Wq, aux3, Cq2, sig2; // second (LQ) preconditioning step
        size_t stop;                     // early stop of the first QR: [ctl (2 doubles) | squared norms of the strips of one step]
        bool lq = false;
        int r0 = 0;                 // rows surviving the up-front deflation
        std::vector<int32_t> good0; // their indices
    };
    std::vector<Lay> lay((size_t)nmat);
    size_t off = 0;
    // all sig arrays first and contiguous: ONE device->host copy after the iteration
    size_t sig_begin = off;
    for (int64_t b = 0; b < nmat; ++b) {
        Lay& l = lay[(size_t)b];
        l.m = (int)sd[b].m;
        l.n = (int)sd[b].n;
        l.k = std::min(l.m, l.n);
        l.L = std::max(l.m, l.n);
        l.kp = round_up(l.k, 64);
        l.tall = l.m >= l.n;
        l.sig = off;
        off += sizeof(double) * (size_t)l.kp;
    }
    for (int64_t b = 0; b < nmat; ++b) { // thresholds right behind the sig arrays: same D2H copy
        lay[(size_t)b].thr = off;
        off += sizeof(double);
    }
    for (int64_t b = 0; b < nmat; ++b) { // ... and the "no left vector" flags of the LQ step
        lay[(size_t)b].bad = off;
        off += sizeof(double);
    }
    const size_t sig_bytes = off - sig_begin;
    off = al(off);
    for (int64_t b = 0; b < nmat; ++b) {
        Lay& l = lay[(size_t)b];
        auto take = [&](size_t bytes) {
            const size_t o = off;
            off += al(bytes);
            return o;
        };
        l.Ac = take(sizeof(double) * (size_t)l.L * l.k);
        l.aux = take(bqr_aux_bytes(l.L, l.k, l.L, l.k));
        l.stop = take(sizeof(double) * (size_t)(2 + (l.k + cyb::NBK - 1) / cyb::NBK * cyb::bqr_strip_slots(l.L)));
        l.W = take(sizeof(double) * (size_t)l.kp * l.kp);
        l.J = take(sizeof(double) * (size_t)l.kp * l.kp);
        l.rank = take(sizeof(int32_t) * (size_t)l.kp);
        l.nnull = take(sizeof(int32_t));
        l.Cq = take(sizeof(double) * (size_t)l.L * l.k);
        l.Fc = take(sizeof(double) * (size_t)l.k * l.k);
        l.aux2 = take(bqr_aux_bytes(l.k, l.k, l.k, l.k));
        l.Cn = take(sizeof(double) * (size_t)l.k * l.k);
        l.Wc = take(sizeof(double) * (size_t)l.kp * l.kp);
        l.Jc = take(sizeof(double) * (size_t)l.kp * l.kp);
        l.Wq = take(sizeof(double) * (size_t)l.kp * l.kp);
        l.aux3 = take(bqr_aux_bytes(l.k, l.k, l.kp, l.k));
        l.Cq2 = take(sizeof(double) * (size_t)l.k * l.k);
        l.sig2 = take(sizeof(double) * (size_t)l.kp);
    }
    // the row-index lists (k entries per matrix) contiguous: ONE device copy fills all of them
    const size_t idx_begin = off;
    for (int64_t b = 0; b < nmat; ++b) {
        lay[(size_t)b].idx = off;
        off += sizeof(int32_t) * (size_t)lay[(size_t)b].k;
    }
    off = al(off);
    void* ws = nullptr;
    CYB_TRY(ctx->workspace(off, &ws, 0));
    char* base = static_cast<char*>(ws);
    CYB_HIP(hipMemsetAsync(ws, 0, off, st));
    auto dp = [&](size_t o) { return reinterpret_cast<double*>(base + o); };

    // ---- 1./2. A -> column-major working copy, blocked Householder QR
    std::vector<BqrMat> qm((size_t)nmat);
    std::vector<XposeDesc> x_in, x_r;
    std::vector<EyeDesc> eyeJ;
    std::vector<JMat> jm((size_t)nmat);
    for (int64_t b = 0; b < nmat; ++b) {
        const Lay& l = lay[(size_t)b];
        BqrMat& q = qm[(size_t)b];
        q.Ac = dp(l.Ac);
        q.ld = l.L;
        q.m = l.L;
        q.n = l.k;
        q.k = l.k;
        bqr_carve(q, base + l.aux, l.k);
        q.v_zeroed = 1; // (the workspace is memset below)
        q.reflect_always = cplx ? 1 : 0;
        // a rank-deficient block (every block of a two-site theta = A.B) stops factoring once the trailing block is at the
        // level the up-front deflation below discards anyway: ||A[j:, j:]||_F <= L eps ||A||_F (numpy.linalg.matrix_rank's
        // threshold; the reference norm the kernels use is a lower bound of ||A||_F, so they stop no earlier than this)
        q.ctl = dp(l.stop);
        q.parts = dp(l.stop) + 2;
        q.stop_rel2 = ((double)l.L * 2.220446049250313e-16) * ((double)l.L * 2.220446049250313e-16);
        if (l.tall) // Ac (col-major m x n) <- A (row-major):  out(r = col, c = row) = A[c*lda + r]
            x_in.push_back(XposeDesc{sd[b].A, q.Ac, sd[b].lda, l.L, l.n, l.m, 0, 0, 0, 0});
        else        // Ac (col-major n x m) = A^T : column c of Ac is row c of A
            x_in.push_back(XposeDesc{sd[b].A, q.Ac, sd[b].lda, l.L, l.m, l.n, 0, 0, 1, 0});
        // W (kp x kp) <- R (k x k upper triangle): out(r, c) = Ac[c*L + r]
        x_r.push_back(XposeDesc{q.Ac, dp(l.W), l.L, l.kp, l.k, l.k, 1, l.k, 0, 0});
        eyeJ.push_back(EyeDesc{dp(l.J), l.kp, l.kp, l.kp, 0, 0});
        JMat& j = jm[(size_t)b];
        j.W = dp(l.W);
        j.J = dp(l.J);
        j.nvp = l.kp;
        j.lenp = l.kp;
        j.nb = row_blocks(l.k, l.kp);
        j.nv = l.k;
        j.len = l.k;
        j.pad = 0;
        j.tol = 2.220446049250313e-16 * std::max(16.0, 4.0 * std::sqrt((double)l.k));
        j.thr2 = 0.0; // set after the row norms of R are known
    }
    CYB_TRY(xpose_batched(ctx, x_in));
    CYB_TRY(bqr_factor(ctx, qm));
    CYB_TRY(xpose_batched(ctx, x_r));
    CYB_TRY(eye_cols_batched(ctx, eyeJ));
    // descriptors of the read-off kernels (also used for the row norms of the up-front deflation)
    std::vector<PostDesc> post((size_t)nmat);
    for (int64_t b = 0; b < nmat; ++b) {
        const Lay& l = lay[(size_t)b];
        PostDesc& q = post[(size_t)b];
        q.W = dp(l.W);
        q.J = dp(l.J);
        q.nvp = l.kp;
        q.lenp = l.kp;
        q.nv = l.k;
        q.len = l.k;
        q.transposed = l.tall ? 0 : 1; // wide: the W side (X) gives the COLUMNS of U
        q.mode = 0;
        q.sig = dp(l.sig);
        q.rank = reinterpret_cast<int32_t*>(base + l.rank);
        q.S = sd[b].S;
        q.U = l.tall ? nullptr : sd[b].U;
        q.ldu = sd[b].ldu;
        q.Vh = l.tall ? sd[b].Vh : nullptr;
        q.ldvh = sd[b].ldvh;
        q.shift = nullptr;
        q.scratch = nullptr;
        q.n_null = reinterpret_cast<int32_t*>(base + l.nnull);
        q.null_scale = 1.0;
        q.cplx = cplx ? 1 : 0;
    }
    void* d_post = nullptr;
    const PostDesc* dpost = nullptr;
    std::vector<double> h_sig(sig_bytes / sizeof(double));
    auto sig_of = [&](const Lay& l) { return h_sig.data() + (l.sig - sig_begin) / sizeof(double); };
    // ---- 2b. up-front deflation: rows of R below the numerical-rank threshold never enter the
    //          iteration (for a rank-deficient block these are the trailing rows of R)
    CYB_TRY(ctx->upload(post.data(), sizeof(PostDesc) * post.size(), &d_post));
    dpost = static_cast<const PostDesc*>(d_post);
    hipLaunchKernelGGL(row_norm_kernel, dim3(helper_grid_x((size_t)nmat), (unsigned)nmat), dim3(256), 0, st, dpost);
    CYB_HIP(hipGetLastError());
    CYB_TRY(ctx->d2h(h_sig.data(), base + sig_begin, sig_bytes));
    {
        std::vector<int32_t> idx_all;
        std::vector<size_t> idx_off((size_t)nmat, 0);
        for (int64_t b = 0; b < nmat; ++b) {
            Lay& l = lay[(size_t)b];
            const double* sg = sig_of(l);
            // numerical-rank threshold (numpy.linalg.matrix_rank's): ||A||_F * max(m,n) * eps, squared;
            // ||A||_F = ||R||_F comes from the row norms just read
            double fro2 = 0.0;
            for (int j = 0; j < l.k; ++j) fro2 += sg[j] * sg[j];
            const double sc = (double)l.L * 2.220446049250313e-16;
            const double thr2 = fro2 * sc * sc;
            jm[(size_t)b].thr2 = thr2;
            std::vector<int32_t> nul;
            for (int j = 0; j < l.k; ++j) {
                const double v = cplx ? std::max(sg[j & ~1], sg[std::min(j | 1, l.k - 1)]) : sg[j];
                (v * v > thr2 ? l.good0 : nul).push_back(j);
            }
            l.r0 = (int)l.good0.size();
            idx_off[(size_t)b] = idx_all.size();
            idx_all.insert(idx_all.end(), l.good0.begin(), l.good0.end());
            idx_all.insert(idx_all.end(), nul.begin(), nul.end());
        }
        // ---- 2c. second preconditioning step (LQ): the surviving rows R_g (r0 x k) are factored once more,
        //          R_g^T = Q2 R2, and the iteration runs on the rows of the r0 x r0 triangle R2 instead of the rows
        //          of R_g.  R2 R2^T is much closer to diagonal than R_g R_g^T (Drmac & Veselic, SIAM J. Matrix Anal.
        //          Appl. 29 (2008), sec. 3: the second QR is what makes one-sided Jacobi converge in a few sweeps):
        //          a theta of a DMRG bond (graded spectrum over 14 decades) needs 8 sweeps instead of 15, and the rows
        //          the rounds stream are r0 long instead of k.  With R2 = J'^T S Z^T (iteration: W' = S Z^T, rotations J')
        //              R_g = Z S (Q2 [J'^T; 0])^T,
        //          so the pair the read-off expects is  W_equiv = S (Q2 [J'^T; 0])^T  and  J_equiv = Z^T, and the
        //          trailing k - r0 columns of Q2 are an orthonormal basis of the complement of R_g's row space: the
        //          completion of the up-front deflated rows needs no QR of its own any more.
        //          ROUND 3: not for every block.  A FULL-RANK, well-conditioned large block (row norms of R within 1e4 of each
        //          other, k >= 128) converges in the same number of sweeps on R itself -- 9-11 against 9-10 -- so the second
        //          factorisation (k / 32 panel steps of ~170 us) is never repaid now that a sweep is cheap: Gaussian 1024^2 26.6 ->
        //          22.9 ms, four 512^2 11.3 -> 9.9 ms (scripts/lq_choice_probe.py).  Rank-deficient blocks keep it (the iteration
        //          shrinks to r0 x r0 and the completion falls out of Q2), graded ones too (fewer sweeps and the better
        //          accuracy of the small values), small ones too (a sweep costs a launch there).  CYB_SVD_LQ_ALWAYS=1: every block.
        static const bool no_lq = getenv("CYB_SVD_NOLQ") != nullptr;
        static const bool lq_always = getenv("CYB_SVD_LQ_ALWAYS") != nullptr;
        static const int lq_skip_k = getenv("CYB_SVD_LQ_SKIP_K") ? atoi(getenv("CYB_SVD_LQ_SKIP_K")) : 128;
        static const double lq_skip_ratio = getenv("CYB_SVD_LQ_SKIP_RATIO") ? atof(getenv("CYB_SVD_LQ_SKIP_RATIO")) : 1e4;
        for (int64_t b = 0; b < nmat; ++b) {
            Lay& l = lay[(size_t)b];
            bool skip = false;
            if (!lq_always && l.r0 == l.k && l.k >= lq_skip_k) {
                const double* sg = sig_of(l);
                double lo = 1e300, hi = 0.0;
                for (int j = 0; j < l.k; ++j) {
                    lo = std::min(lo, sg[j]);
                    hi = std::max(hi, sg[j]);
                }
                skip = hi <= lq_skip_ratio * lo;
            }
            l.lq = !no_lq && l.r0 >= 2 && !skip;
        }
        void* d_idx_v = nullptr;
        CYB_TRY(ctx->upload(idx_all.data(), sizeof(int32_t) * idx_all.size(), &d_idx_v));
        // (k entries per matrix, in the order of the lists in the workspace)
        CYB_HIP(hipMemcpyAsync(base + idx_begin, d_idx_v, sizeof(int32_t) * idx_all.size(), hipMemcpyDeviceToDevice, st));
    }
    const std::vector<JMat> jm0 = jm;
    std::vector<BqrMat> qm3; // the LQ factorisations (their Q2 also completes the deflated rows in step 5)
    std::vector<int> qm3_of((size_t)nmat, -1);
    std::vector<int32_t> sweeps;
    auto iterate = [&](bool allow_lq, int& jst) -> int {
        jm = jm0;
        qm3.clear();
        std::fill(qm3_of.begin(), qm3_of.end(), -1);
        std::vector<RowMoveDesc> gat, zer;
        std::vector<EyeDesc> eyeJc;
        std::vector<XposeDesc> x_r2;
        std::vector<LqDesc> lqd;
        std::vector<BqrTarget> tg3;
        for (int64_t b = 0; b < nmat; ++b) {
            Lay& l = lay[(size_t)b];
            l.lq = l.lq && allow_lq;
            if (l.r0 == l.k && !l.lq) continue;
            const int32_t* didx = reinterpret_cast<const int32_t*>(base + l.idx);
            const int rp = std::max(round_up(l.r0, 64), 64);
            gat.push_back(RowMoveDesc{dp(l.W), dp(l.Wc), didx, l.kp, l.kp, l.r0, l.kp, 0, 0});
            if (l.r0 < l.k) zer.push_back(RowMoveDesc{nullptr, dp(l.W), didx + l.r0, 0, l.kp, l.k - l.r0, l.kp, 2, 0});
            eyeJc.push_back(EyeDesc{dp(l.Jc), rp, rp, rp, 0, 0});
            JMat& j = jm[(size_t)b];
            j.W = dp(l.Wc);
            j.J = dp(l.Jc);
            j.nvp = rp;
            j.nb = row_blocks(l.r0, rp);
            j.nv = l.r0;
            if (l.lq) {
                // Wc (r0 x kp row-major) read column-major with ld = kp IS R_g^T (k x r0): factored in place
                BqrMat q;
                q.Ac = dp(l.Wc);
                q.ld = l.kp;
                q.m = l.k;
                q.n = l.r0;
                q.k = l.r0;
                bqr_carve(q, base + l.aux3, l.k);
                q.v_zeroed = allow_lq ? 1 : 0; // (first use of aux3 after the memset of the workspace)
                q.reflect_always = cplx ? 1 : 0;
                // Wq (rp x rp row-major) <- R2 (upper triangle of the factored Wc)
                x_r2.push_back(XposeDesc{dp(l.Wc), dp(l.Wq), l.kp, rp, l.r0, l.r0, 1, l.r0, 0, 0});
                j.W = dp(l.Wq);
                j.lenp = rp;
                j.len = l.r0;
                const double thr2 = j.thr2;
                j.thr2 = 0.0; // no deflation inside the iteration: every row of S Z^T must keep its left vector
                lqd.push_back(LqDesc{dp(l.Wq), dp(l.Jc), dp(l.Wc), dp(l.Cq2), dp(l.sig2), dp(l.bad), thr2, rp, l.kp, l.k, l.r0, cplx ? 1 : 0, 0});
                tg3.push_back(BqrTarget{(int)qm3.size(), dp(l.Cq2), l.k, l.k});
                qm3_of[(size_t)b] = (int)qm3.size();
                qm3.push_back(q);
            }
        }
        CYB_TRY(launch_row_moves(ctx, gat));
        CYB_TRY(launch_row_moves(ctx, zer));
        CYB_TRY(eye_cols_batched(ctx, eyeJc));
        if (!qm3.empty()) {
            CYB_TRY(bqr_factor(ctx, qm3));
            CYB_TRY(xpose_batched(ctx, x_r2));
        }
        if (cplx) {
            std::vector<PairDesc> prs;
            for (const JMat& j : jm)
                if (j.nv >= 2) prs.push_back(PairDesc{const_cast<double*>(j.W), j.lenp, j.nv & ~1, j.len & ~1});
            if (!prs.empty()) {
                void* d = nullptr;
                CYB_TRY(ctx->upload(prs.data(), sizeof(PairDesc) * prs.size(), &d));
                hipLaunchKernelGGL(embed_pairs_kernel, dim3(64, (unsigned)prs.size()), dim3(256), 0, st, static_cast<const PairDesc*>(d));
                CYB_HIP(hipGetLastError());
            }
        }
        // ---- 3. block Jacobi (plain mode: threshold deflation stays on for rows that fall below it later)
        jst = jacobi_orthogonalise(ctx, jm, 40, sweeps, cplx);
        if (jst != CYB_OK && jst != CYB_ERR_NOCONV) return jst;
        if (!qm3.empty()) {
            void* d = nullptr;
            CYB_TRY(ctx->upload(lqd.data(), sizeof(LqDesc) * lqd.size(), &d));
            const LqDesc* dl = static_cast<const LqDesc*>(d);
            hipLaunchKernelGGL(lq_pack_kernel, dim3(64, (unsigned)lqd.size()), dim3(256), 0, st, dl);
            hipLaunchKernelGGL(lq_rows_kernel, dim3(64, (unsigned)lqd.size()), dim3(256), 0, st, dl);
            // (its waves share the Gram rows of the flagged directions: a large graded block has hundreds -- 0.94 ms on 64 workgroups)
            hipLaunchKernelGGL(lq_check_kernel, dim3(helper_grid_x(lqd.size()), (unsigned)lqd.size()), dim3(256), 0, st, dl);
            CYB_HIP(hipGetLastError());
            CYB_TRY(bqr_apply_q(ctx, qm3, tg3));
            CYB_TRY(ctx->upload(lqd.data(), sizeof(LqDesc) * lqd.size(), &d)); // (slot may have been recycled)
            hipLaunchKernelGGL(lq_unpack_kernel, dim3(64, (unsigned)lqd.size()), dim3(256), 0, st, static_cast<const LqDesc*>(d));
            CYB_HIP(hipGetLastError());
        }
        return CYB_OK;
    };
    int jst = CYB_OK;
    CYB_TRY(iterate(true, jst));
    if (!qm3.empty()) {
        // a row of S Z^T that is exactly zero has no left vector to read off, and rows of pure rounding noise can
        // keep the iteration from settling: both are cases for the plain iteration (deflation on), from the rows of
        // R that W still holds
        std::vector<double> bad((size_t)nmat, 0.0);
        CYB_TRY(ctx->d2h(bad.data(), base + lay[0].bad, sizeof(double) * (size_t)nmat));
        static const bool force_redo = getenv("CYB_SVD_LQ_FORCE_REDO") != nullptr; // (test hook: exercise the fallback)
        bool redo = jst == CYB_ERR_NOCONV || force_redo;
        for (double v : bad) redo = redo || v != 0.0;
        static const bool trace_redo = getenv("CYB_SVD_TRACE_REDO") != nullptr;
        if (redo && trace_redo) {
            int nb = 0;
            for (double v : bad) nb += v != 0.0;
            fprintf(stderr, "[cyb] svd: plain iteration after the LQ one (%d of %lld blocks flagged, noconv %d)\n", nb, (long long)nmat,
                    (int)(jst == CYB_ERR_NOCONV));
        }
        if (redo) CYB_TRY(iterate(false, jst));
    }
    if (info)
        for (int64_t b = 0; b < nmat; ++b) info[b] = sweeps[(size_t)b];
    if (jst != CYB_OK && jst != CYB_ERR_NOCONV) return jst;
    {   // compact results back into the full-size W / J (J is still the identity there)
        std::vector<RowMoveDesc> sw, sj;
        for (int64_t b = 0; b < nmat; ++b) {
            const Lay& l = lay[(size_t)b];
            if ((l.r0 == l.k && !l.lq) || l.r0 == 0) continue;
            const int32_t* didx = reinterpret_cast<const int32_t*>(base + l.idx);
            const int rp = std::max(round_up(l.r0, 64), 64);
            sw.push_back(RowMoveDesc{dp(l.Wc), dp(l.W), didx, l.kp, l.kp, l.r0, l.kp, 1, 0});
            sj.push_back(RowMoveDesc{dp(l.Jc), dp(l.J), didx, rp, l.kp, l.r0, l.r0, 0, 0});
        }
        CYB_TRY(launch_row_moves(ctx, sw));
        CYB_TRY(launch_row_moves(ctx, sj, true));
    }
    // ---- 4. singular values (row norms) -> host, to find the deflated rows
    CYB_TRY(ctx->upload(post.data(), sizeof(PostDesc) * post.size(), &d_post));
    dpost = static_cast<const PostDesc*>(d_post);
    hipLaunchKernelGGL(row_norm_kernel, dim3(helper_grid_x((size_t)nmat), (unsigned)nmat), dim3(256), 0, st, dpost);
    CYB_HIP(hipGetLastError());
    CYB_TRY(ctx->d2h(h_sig.data(), base + sig_begin, sig_bytes));
    // ---- 5. orthonormal completion of the deflated rows from a full Householder Q
    {
        std::vector<int32_t> idx_all;
        struct Comp {
            int64_t b;
            size_t good_off, null_off;
            int r, q;
        };
        std::vector<Comp> comps;
        for (int64_t b = 0; b < nmat; ++b) {
            const Lay& l = lay[(size_t)b];
            const double* sg = sig_of(l);
            std::vector<int32_t> good, nul;
            for (int j = 0; j < l.k; ++j) {
                const double v = cplx ? std::max(sg[j & ~1], sg[std::min(j | 1, l.k - 1)]) : sg[j];
                (v > 0.0 ? good : nul).push_back(j);
            }
            if (rank_out) rank_out[b] = (int32_t)good.size();
            Comp c{b, idx_all.size(), 0, (int)good.size(), (int)nul.size()};
            idx_all.insert(idx_all.end(), good.begin(), good.end()); // (every matrix: the lists are rewritten in one copy)
            c.null_off = idx_all.size();
            idx_all.insert(idx_all.end(), nul.begin(), nul.end());
            // CYB_SVD_SKIP_NULL_VECTORS: the caller will discard the singular vectors of the deflated (numerically zero)
            // singular values -- a truncated SVD keeps the chi_max largest -- so their orthonormal completion is not
            // computed; those rows of the W-side factor stay zero
            if (nul.empty() || (flags & CYB_SVD_SKIP_NULL_VECTORS)) continue;
            comps.push_back(c);
        }
        if (!comps.empty()) {
            // the index lists must outlive many later uploads: stage through the ring, keep in the workspace
            void* d_idx_v = nullptr;
            CYB_TRY(ctx->upload(idx_all.data(), sizeof(int32_t) * idx_all.size(), &d_idx_v));
            CYB_HIP(hipMemcpyAsync(base + idx_begin, d_idx_v, sizeof(int32_t) * idx_all.size(), hipMemcpyDeviceToDevice, st));
            std::vector<RowsDesc> gath, scat;
            std::vector<int32_t> cols_all;                    // LQ mode: column of Cq2 for every null row ...
            std::vector<std::pair<size_t, size_t>> lq_scat;   // ... (index into scat, offset into cols_all)
            std::vector<BqrMat> qm2;
            std::vector<BqrTarget> tg2;
            std::vector<EyeDesc> eyes;
            for (const Comp& c : comps) {
                const Lay& l = lay[(size_t)c.b];
                if (l.lq) {
                    // every null row has its unit vector in Cq2 = Q2 [J'^T 0; 0 I] already (step 2c): a row deflated up
                    // front takes one of the trailing columns (the complement of R_g's row space), a row the read-off
                    // of the LQ iteration found numerically null (sigma_i^2 <= thr2) its own column i -- no QR
                    const size_t c0 = cols_all.size();
                    int t = 0;
                    size_t g = 0;
                    const int32_t* nulp = idx_all.data() + c.null_off;
                    for (int qn = 0; qn < c.q; ++qn) {
                        const int32_t j = nulp[qn];
                        while (g < l.good0.size() && l.good0[g] < j) ++g;
                        if (g < l.good0.size() && l.good0[g] == j) cols_all.push_back((int32_t)g);
                        else cols_all.push_back((int32_t)(l.r0 + t++));
                    }
                    lq_scat.push_back({scat.size(), c0});
                    scat.push_back(RowsDesc{dp(l.W), dp(l.sig), reinterpret_cast<const int32_t*>(base + l.idx) + c.r, dp(l.Cq2), l.kp,
                                            l.k, c.q, 0, nullptr});
                    continue;
                }
                if (c.r > 0) {
                    gath.push_back(RowsDesc{dp(l.W), dp(l.sig), reinterpret_cast<const int32_t*>(base + l.idx), dp(l.Fc), l.kp, l.k, c.r, 0, nullptr});
                    BqrMat q;
                    q.Ac = dp(l.Fc);
                    q.ld = l.k;
                    q.m = l.k;
                    q.n = c.r;
                    q.k = std::min(l.k, c.r);
                    bqr_carve(q, base + l.aux2, c.q);
                    tg2.push_back(BqrTarget{(int)qm2.size(), dp(l.Cn), l.k, c.q});
                    qm2.push_back(q);
                }
                // Cn (k x q) = columns r .. r+q-1 of the identity (r = 0: the completion is the identity itself)
                eyes.push_back(EyeDesc{dp(l.Cn), l.k, l.k, c.q, c.r, 0});
                scat.push_back(RowsDesc{dp(l.W), dp(l.sig), reinterpret_cast<const int32_t*>(base + l.idx) + c.r, dp(l.Cn), l.kp, l.k, c.q, 0, nullptr});
            }
            if (!gath.empty()) {
                void* d = nullptr;
                CYB_TRY(ctx->upload(gath.data(), sizeof(RowsDesc) * gath.size(), &d));
                hipLaunchKernelGGL(gather_rows_kernel, dim3(64, (unsigned)gath.size()), dim3(256), 0, st,
                                   static_cast<const RowsDesc*>(d));
                CYB_HIP(hipGetLastError());
                CYB_TRY(bqr_factor(ctx, qm2));
            }
            CYB_TRY(eye_cols_batched(ctx, eyes));
            if (!tg2.empty()) CYB_TRY(bqr_apply_q(ctx, qm2, tg2));
            void* d = nullptr;
            if (!cols_all.empty()) {
                void* d_cols = nullptr;
                CYB_TRY(ctx->upload(cols_all.data(), sizeof(int32_t) * cols_all.size(), &d_cols));
                for (const auto& pr : lq_scat) scat[pr.first].col = static_cast<const int32_t*>(d_cols) + pr.second;
            }
            CYB_TRY(ctx->upload(scat.data(), sizeof(RowsDesc) * scat.size(), &d));
            hipLaunchKernelGGL(scatter_rows_kernel, dim3(64, (unsigned)scat.size()), dim3(256), 0, st,
                               static_cast<const RowsDesc*>(d));
            CYB_HIP(hipGetLastError());
        }
    }
    // ---- 6. sort, write the W side directly and the J side through the block reflectors of Q1
    CYB_TRY(ctx->upload(post.data(), sizeof(PostDesc) * post.size(), &d_post)); // (slot may have been recycled)
    dpost = static_cast<const PostDesc*>(d_post);
    hipLaunchKernelGGL(rank_kernel, dim3(8, (unsigned)nmat), dim3(256), 0, st, dpost);
    hipLaunchKernelGGL(write_factors_kernel, dim3(helper_grid_x((size_t)nmat), (unsigned)nmat), dim3(256), 0, st, dpost);
    CYB_HIP(hipGetLastError());
    std::vector<JcqDesc> jc;
    std::vector<BqrTarget> tg;
    std::vector<XposeDesc> x_out;
    for (int64_t b = 0; b < nmat; ++b) {
        const Lay& l = lay[(size_t)b];
        jc.push_back(JcqDesc{dp(l.J), reinterpret_cast<const int32_t*>(base + l.rank), dp(l.Cq), l.L, l.kp, l.k});
        tg.push_back(BqrTarget{(int)b, dp(l.Cq), l.L, l.k});
        if (l.tall) // U (row-major m x k): out(r, c) = Cq[c*L + r]
            x_out.push_back(XposeDesc{dp(l.Cq), sd[b].U, l.L, sd[b].ldu, l.m, l.k, 0, 0, 0, 0});
        else        // Vh (row-major k x n): row r = column r of Cq
            x_out.push_back(XposeDesc{dp(l.Cq), sd[b].Vh, l.L, sd[b].ldvh, l.k, l.n, 0, 0, 1, 0});
    }
    {
        void* d = nullptr;
        CYB_TRY(ctx->upload(jc.data(), sizeof(JcqDesc) * jc.size(), &d));
        hipLaunchKernelGGL(j_to_cq_kernel, dim3(helper_grid_x((size_t)nmat), (unsigned)nmat), dim3(256), 0, st, static_cast<const JcqDesc*>(d));
        CYB_HIP(hipGetLastError());
    }
    CYB_TRY(bqr_apply_q(ctx, qm, tg));
    CYB_TRY(xpose_batched(ctx, x_out));
    if (info) CYB_HIP(hipStreamSynchronize(st));
    return jst;
}

} // namespace cyb

namespace cyb {
bool svd_small_fits(int64_t m, int64_t n);
int svd_small_batched(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* sweeps_out);
int eigh_small_batched(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* sweeps_out);
} // namespace cyb

extern "C" {

static int svd_batched_impl(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* info, int flags = 0,
                            int32_t* rank_out = nullptr)
{
    CYB_REQUIRE(ctx, "cyb_svd_batched_f64: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_svd_batched_f64: bad descriptor list");
    // empty blocks need no work; the engine pads everything else
    std::vector<cyb_svd_desc> nz;
    std::vector<int64_t> idx;
    for (int64_t b = 0; b < n; ++b) {
        if (info) info[b] = 0;
        if (rank_out) rank_out[b] = (int32_t)std::min(descs[b].m, descs[b].n); // (paths that always complete report k)
        if (descs[b].m > 0 && descs[b].n > 0) {
            CYB_REQUIRE(descs[b].U && descs[b].Vh, "svd block %lld: U / Vh is NULL", (long long)b);
            nz.push_back(descs[b]);
            idx.push_back(b);
        }
    }
    // small blocks: direct Jacobi; larger ones: QR-preconditioned pipeline
    static const bool no_qr = getenv("CYB_SVD_NOQR") != nullptr;
    // tiny blocks (min <= 64, max <= 128): the whole iteration in LDS, one workgroup per block (svd_small.hip)
    static const bool no_small = getenv("CYB_SVD_NOSMALL") != nullptr;
    // ... when they are the whole list or many: a few tiny sectors next to large ones (the two or three smallest blocks
    // of a chi=4096 theta) ride along in the rounds of the large blocks for free, whereas the fused kernel would sit in
    // front of the pipeline on the same stream (0.66 ms of a 53 ms call)
    size_t n_fit = 0;
    for (const auto& d : nz) n_fit += cyb::svd_small_fits(d.m, d.n) ? 1 : 0;
    const bool embedded = (flags & CYB_SVD_EMBEDDED_COMPLEX) != 0; // (complex blocks as real embeddings: one pipeline for every size)
    const bool use_small = !embedded && !no_small && (n_fit == nz.size() || n_fit >= 16);
    std::vector<cyb_svd_desc> tiny, small, large;
    std::vector<int64_t> idx_t, idx_s, idx_l;
    // blocks with min(m, n) >= 48 go through the QR-preconditioned pipeline; once there is one, the smaller blocks of the
    // list join it (their panels and sweeps ride along in the same launches) instead of forming a second, serial
    // iteration behind it -- a DMRG bond at chi = 256 has sectors from 4 x 4 to 140 x 140 in one call
    static const bool no_merge = getenv("CYB_SVD_NOMERGE") != nullptr;
    bool any_large = false;
    for (const auto& d : nz) any_large = any_large || (!no_qr && std::min(d.m, d.n) >= 48 && !(use_small && cyb::svd_small_fits(d.m, d.n)));
    // A block that does not fit the in-LDS kernel goes through the pipeline as well, whatever its size: the direct iteration
    // (run_jacobi on the rows of A) has no deflation of numerically zero rows, and a rank-deficient 38 x 135 block with 105
    // zero columns kept rotating rounding noise for 40 sweeps (scripts/tensor_fuzz.py, seed 11; the up-front deflation of
    // the pipeline takes those rows out before the first sweep).  CYB_SVD_NOMERGE keeps the old split for the tests.
    (void)any_large;
    const int64_t large_min = embedded ? 1 : no_merge ? 48 : 2;
    for (size_t k = 0; k < nz.size(); ++k) {
        if (use_small && cyb::svd_small_fits(nz[k].m, nz[k].n)) {
            CYB_REQUIRE(nz[k].S, "svd block %lld: S is NULL", (long long)idx[k]);
            tiny.push_back(nz[k]);
            idx_t.push_back(idx[k]);
        } else if (!no_qr && std::min(nz[k].m, nz[k].n) >= large_min) {
            large.push_back(nz[k]);
            idx_l.push_back(idx[k]);
        } else {
            small.push_back(nz[k]);
            idx_s.push_back(idx[k]);
        }
    }
    std::vector<int32_t> inf_t(tiny.size()), inf_s(small.size()), inf_l(large.size());
    const int st_t = cyb::svd_small_batched(ctx, tiny.data(), (int64_t)tiny.size(), inf_t.data());
    if (st_t != CYB_OK && st_t != CYB_ERR_NOCONV) return st_t;
    if (info)
        for (size_t k = 0; k < tiny.size(); ++k) info[idx_t[k]] = inf_t[k];
    if (small.empty() && large.empty()) return st_t;
    std::vector<int32_t> rank_l(large.size());
    const int st_l = cyb::run_svd_qr(ctx, (int64_t)large.size(), large.data(), info ? inf_l.data() : nullptr, flags,
                                     rank_out ? rank_l.data() : nullptr);
    if (rank_out)
        for (size_t k = 0; k < large.size(); ++k) rank_out[idx_l[k]] = rank_l[k];
    if (st_l != CYB_OK && st_l != CYB_ERR_NOCONV) return st_l;
    const int st_s = cyb::run_jacobi(ctx, 0, (int64_t)small.size(), small.data(), nullptr, info ? inf_s.data() : nullptr);
    if (info) {
        for (size_t k = 0; k < small.size(); ++k) info[idx_s[k]] = inf_s[k];
        for (size_t k = 0; k < large.size(); ++k) info[idx_l[k]] = inf_l[k];
    }
    return st_s != CYB_OK ? st_s : (st_l != CYB_OK ? st_l : st_t);
}

static int eigh_batched_impl(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* info, int flags = 0)
{
    CYB_REQUIRE(ctx, "cyb_eigh_batched_f64: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_eigh_batched_f64: bad descriptor list");
    std::vector<cyb_eigh_desc> nz;
    std::vector<int64_t> idx;
    for (int64_t b = 0; b < n; ++b) {
        if (info) info[b] = 0;
        if (descs[b].n > 0) {
            nz.push_back(descs[b]);
            idx.push_back(b);
        }
    }
    std::vector<int32_t> inf(nz.size());
    // lists made of small blocks only (n <= 64): the fused in-LDS kernel of svd_small.hip
    static const bool no_small = getenv("CYB_SVD_NOSMALL") != nullptr;
    const bool embedded = (flags & CYB_EIGH_EMBEDDED_COMPLEX) != 0;
    bool all_small = !no_small && !nz.empty() && !embedded;
    for (const auto& d : nz) all_small = all_small && d.n <= 64;
    const int st = all_small ? cyb::eigh_small_batched(ctx, nz.data(), (int64_t)nz.size(), inf.data())
                             : cyb::run_jacobi(ctx, 1, (int64_t)nz.size(), nullptr, nz.data(), info ? inf.data() : nullptr, embedded);
    if (info)
        for (size_t k = 0; k < nz.size(); ++k) info[idx[k]] = inf[k];
    return st;
}

// ---- range-safe entry points (scaling.hip): blocks whose entries sit outside [1e-90, 1e90] are decomposed as
//      s*A (s a power of two) and the scale is taken out of the singular values / eigenvalues afterwards
static int svd_batched_ranged(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* info, int flags, int32_t* rank)
{
    std::vector<cyb::MatRef> refs;
    std::vector<int64_t> which;
    for (int64_t b = 0; b < n; ++b)
        if (descs[b].m > 0 && descs[b].n > 0 && descs[b].A) {
            refs.push_back(cyb::MatRef{descs[b].A, descs[b].lda, descs[b].m, descs[b].n});
            which.push_back(b);
        }
    std::vector<double> amax;
    CYB_TRY(cyb::matrix_amax(ctx, refs, amax));
    std::vector<cyb_svd_desc> mod;
    std::vector<void*> temps;
    std::vector<cyb::ScaleJob> pre, post;
    for (size_t k = 0; k < refs.size(); ++k) {
        const double sc = cyb::range_scale(amax[k]);
        if (sc == 1.0) continue;
        if (mod.empty()) mod.assign(descs, descs + n);
        cyb_svd_desc& d = mod[(size_t)which[k]];
        void* t = nullptr;
        CYB_HIP(hipMalloc(&t, sizeof(double) * (size_t)d.m * (size_t)d.n));
        temps.push_back(t);
        pre.push_back(cyb::ScaleJob{d.A, d.lda, static_cast<double*>(t), d.n, d.m, d.n, sc});
        d.A = static_cast<const double*>(t);
        d.lda = d.n;
        post.push_back(cyb::ScaleJob{d.S, 1, d.S, 1, std::min(d.m, d.n), 1, 1.0 / sc});
    }
    if (mod.empty()) return svd_batched_impl(ctx, descs, n, info, flags, rank);
    int st = cyb::scale_copy_batched(ctx, pre);
    if (st == CYB_OK) st = svd_batched_impl(ctx, mod.data(), n, info, flags, rank);
    if (st == CYB_OK || st == CYB_ERR_NOCONV) {
        const int st2 = cyb::scale_copy_batched(ctx, post);
        if (st2 != CYB_OK) st = st2;
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (void* t : temps) (void)hipFree(t);
    return st;
}

int cyb_svd_batched_f64(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* info)
{
    CYB_REQUIRE(ctx, "cyb_svd_batched_f64: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_svd_batched_f64: bad descriptor list");
    return svd_batched_ranged(ctx, descs, n, info, 0, nullptr);
}

int cyb_svd_batched_ex_f64(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* info, int32_t flags, int32_t* rank)
{
    CYB_REQUIRE(ctx, "cyb_svd_batched_ex_f64: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_svd_batched_ex_f64: bad descriptor list");
    CYB_REQUIRE((flags & ~(CYB_SVD_SKIP_NULL_VECTORS | CYB_SVD_EMBEDDED_COMPLEX)) == 0, "cyb_svd_batched_ex_f64: unknown flag bits 0x%x", flags);
    if (flags & CYB_SVD_EMBEDDED_COMPLEX)
        for (int64_t b = 0; b < n; ++b)
            CYB_REQUIRE(descs[b].m % 2 == 0 && descs[b].n % 2 == 0, "svd block %lld: an embedded complex block has even extents", (long long)b);
    // (blocks outside the safe exponent range are decomposed as scaled copies, with the same flags: the scale is a power
    //  of two, the rank threshold is relative)
    const int st = svd_batched_ranged(ctx, descs, n, info, flags, rank);
    if (rank && !info) CYB_HIP(hipStreamSynchronize(ctx->stream)); // (the ranks are host data read during the call)
    return st;
}

static int eigh_batched_ranged(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* info, int flags)
{
    std::vector<cyb::MatRef> refs;
    std::vector<int64_t> which;
    for (int64_t b = 0; b < n; ++b)
        if (descs[b].n > 0 && descs[b].A) {
            refs.push_back(cyb::MatRef{descs[b].A, descs[b].lda, descs[b].n, descs[b].n});
            which.push_back(b);
        }
    std::vector<double> amax;
    CYB_TRY(cyb::matrix_amax(ctx, refs, amax));
    std::vector<cyb_eigh_desc> mod;
    std::vector<void*> temps;
    std::vector<cyb::ScaleJob> pre, post;
    for (size_t k = 0; k < refs.size(); ++k) {
        const double sc = cyb::range_scale(amax[k]);
        if (sc == 1.0) continue;
        if (mod.empty()) mod.assign(descs, descs + n);
        cyb_eigh_desc& d = mod[(size_t)which[k]];
        void* t = nullptr;
        CYB_HIP(hipMalloc(&t, sizeof(double) * (size_t)d.n * (size_t)d.n));
        temps.push_back(t);
        pre.push_back(cyb::ScaleJob{d.A, d.lda, static_cast<double*>(t), d.n, d.n, d.n, sc});
        d.A = static_cast<const double*>(t);
        d.lda = d.n;
        post.push_back(cyb::ScaleJob{d.W, 1, d.W, 1, d.n, 1, 1.0 / sc});
    }
    if (mod.empty()) return eigh_batched_impl(ctx, descs, n, info, flags);
    int st = cyb::scale_copy_batched(ctx, pre);
    if (st == CYB_OK) st = eigh_batched_impl(ctx, mod.data(), n, info, flags);
    if (st == CYB_OK || st == CYB_ERR_NOCONV) {
        const int st2 = cyb::scale_copy_batched(ctx, post);
        if (st2 != CYB_OK) st = st2;
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (void* t : temps) (void)hipFree(t);
    return st;
}

int cyb_eigh_batched_f64(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* info)
{
    CYB_REQUIRE(ctx, "cyb_eigh_batched_f64: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_eigh_batched_f64: bad descriptor list");
    return eigh_batched_ranged(ctx, descs, n, info, 0);
}

int cyb_eigh_batched_ex_f64(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* info, int32_t flags)
{
    CYB_REQUIRE(ctx, "cyb_eigh_batched_ex_f64: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_eigh_batched_ex_f64: bad descriptor list");
    CYB_REQUIRE((flags & ~CYB_EIGH_EMBEDDED_COMPLEX) == 0, "cyb_eigh_batched_ex_f64: unknown flag bits 0x%x", flags);
    if (flags & CYB_EIGH_EMBEDDED_COMPLEX)
        for (int64_t b = 0; b < n; ++b)
            CYB_REQUIRE(descs[b].n % 2 == 0, "eigh block %lld: an embedded complex block has an even extent", (long long)b);
    return eigh_batched_ranged(ctx, descs, n, info, flags);
}

} // extern "C"
