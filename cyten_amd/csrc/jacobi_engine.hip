// One-sided block-Jacobi orthogonalisation on gfx950 (engine of the batched SVD / eigh).
//
// Replaces the per-block LAPACK calls of the reference (scipy.linalg.svd -> dgesdd,
// src/block_backend/numpy.cpp:1247-1297; np.linalg.eigh -> dsyevd, numpy.cpp:658-680) with a
// Hestenes one-sided Jacobi in *block* form so that the O(len * nv^2) work per sweep runs on the
// f64 MFMA pipe:
//   for every round of the round-robin schedule over blocks of JB=32 vectors, every workgroup
//   owns one block pair (P,Q) of one matrix and does
//     1. Gram   G = X X^T            X = [W_P; W_Q]  (64 x len), MFMA 16x16x4 f64, K = len
//     2. eigh   G = Qm L Qm^T        two-sided Jacobi on the 64x64 Gram matrix held in LDS:
//                                    all 32 disjoint rotations of a round are applied in ONE pass
//                                    over 2x2 sub-blocks (rows and columns at once)
//     3. update X <- Qm^T X, J_PQ <- Qm^T J_PQ      MFMA again (M=64, K=64, N=len)
//   Rounds are separate launches (the next round needs this round's rows); the per-sweep
//   convergence measure max |g_ij|/sqrt(g_ii g_jj) is accumulated with an atomic max and read by
//   the host once per sweep.
#include "jacobi_engine.h"

#include <algorithm>
#include <cstdlib>

namespace cyb {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define GLOBAL_AS __attribute__((address_space(1)))
typedef const GLOBAL_AS d2* gc2;
typedef const GLOBAL_AS d2* gcp2;
constexpr int PD = 4; // prefetch depth (tiles in flight per thread) of the Gram and update loops
typedef GLOBAL_AS double* gp;

constexpr int GS = JP + 2;   // row stride of the Gram matrix in LDS
constexpr int QS = JP + 16;  // row stride of Qm in LDS ([k][m] layout: 48 = 16 mod 32 -> conflict-free)
constexpr int CS = 64 + 16;  // row stride of the update chunk in LDS ([k][n] layout, 64 columns)
constexpr int GK = 64;       // k extent of one Gram staging tile
constexpr int XS = GK + 2;   // its row stride ([m][k] layout)
constexpr int NT = 256;
constexpr int NPAIR = JP / 2;
static_assert(JP == 32, "the wave tiling below assumes 32 x 32 pair problems");

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// max over the workgroup; `red` is 8 doubles of LDS. All threads get the result.
__device__ __forceinline__ double block_max(double v, double* red, int tid)
{
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// scaled off-diagonal measure of the JP x JP Gram matrix in LDS
__device__ __forceinline__ double gram_offmax(const double* Gs, double* red, int tid)
{
    double m = 0.0;
#pragma unroll
    for (int e = tid; e < JP * JP; e += NT) {
        const int i = e / JP, j = e % JP;
        if (i < j) {
            const double den = Gs[i * GS + i] * Gs[j * GS + j];
            if (den > 0.0) m = fmax(m, fabs(Gs[i * GS + j]) * rsqrt(den));
        }
    }
    return block_max(m, red, tid);
}

// pair of players meeting in round r (0..n-2), slot k (0..n/2-1) of the circle method, n even
__host__ __device__ __forceinline__ void circle_pair(int n, int r, int k, int& p, int& q)
{
    const int m = n - 1;
    if (k == 0) {
        p = m;
        q = r;
    } else {
        p = r + k;
        if (p >= m) p -= m;
        q = r - k;
        if (q < 0) q += m;
    }
}

// symmetric 2x2 Jacobi rotation [[c, s], [-s, c]] that annihilates b in [[a, b], [b, d]]
__device__ __forceinline__ void jacobi_rot(double a, double d, double b, double& c, double& s)
{
    c = 1.0;
    s = 0.0;
    if (fabs(b) > 1e-300) {
        const double tau = (d - a) / (2.0 * b);
        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        c = rsqrt(1.0 + t * t);
        s = t * c;
    }
}

__device__ __forceinline__ int xrow(int i, int P, int Q) { return (i < JB) ? P * JB + i : Q * JB + (i - JB); }

// out(JP x ncols) = Qm^T X for the JP rows {P-block, Q-block} of the row-major matrix `base`
// (row stride ld, ncols a multiple of 64), in place.  Each wave owns one 16-column tile of the
// 64-column chunk and both 16-row tiles.
// Output row r takes eigenvector column perm[r] (descending-norm order inside the pair); rows
// with zrow[perm[r]] != 0 are written as zeros (deflated rows; only when `zero_null`).
__device__ __forceinline__ void apply_update(double* __restrict__ base_, int ld, int ncols, int P, int Q,
                                             const double* Qs, double* Xc, int tid, const int* perm, const int* zrow,
                                             bool zero_null)
{
    const int lane = tid & 63, wave = tid >> 6;
    gp base = (gp)base_;
    // chunk: 32 rows x 64 cols = 1024 d2 / 256 threads = 4 each; v -> (row = v>>5, cv = v&31).
    // PD chunks are kept in flight in registers (the loop is latency bound otherwise: one workgroup
    // per CU, ~1 us per dependent HBM/MALL access).
    d2 reg[PD][4];
    gcp2 src[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int v = tid + p * NT;
        src[p] = (gcp2)(base + (int64_t)xrow(v >> 5, P, Q) * ld + 2 * (v & 31));
    }
    const int nchunk = ncols / 64;
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < nchunk) {
#pragma unroll
            for (int p = 0; p < 4; ++p) reg[d][p] = src[p][d * 32];
        }
    const int m0 = perm[lane & 15], m1 = perm[16 + (lane & 15)];
    const double* ap0 = Qs + (lane >> 4) * QS + m0;
    const double* ap1 = Qs + (lane >> 4) * QS + m1;
    const double* bp = Xc + (lane >> 4) * CS + wave * 16 + (lane & 15);
    int zr[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) zr[i][r] = zero_null ? zrow[perm[i * 16 + (lane >> 4) + 4 * r]] : 0;
    for (int c0 = 0; c0 < nchunk; c0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int c = c0 + d;
            if (c < nchunk) {
                __syncthreads(); // previous chunk's LDS reads are done
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int v = tid + p * NT;
                    *reinterpret_cast<d2*>(Xc + (v >> 5) * CS + 2 * (v & 31)) = reg[d][p];
                }
                __syncthreads();
                if (c + PD < nchunk) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) reg[d][p] = src[p][(c + PD) * 32];
                }
                d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
                // A[m][k] = Qm[k][perm[m]]  (Qs is [k][m]);  B[k][n] = Xc[k][n]
#pragma unroll
                for (int kk = 0; kk < JP / 4; ++kk) {
                    const double b = bp[kk * 4 * CS];
                    const double a0 = ap0[kk * 4 * QS], a1 = ap1[kk * 4 * QS];
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i * 16 + (lane >> 4) + 4 * r;
                        const double val = zr[i][r] ? 0.0 : acc[i][r];
                        base[(int64_t)xrow(row, P, Q) * ld + c * 64 + wave * 16 + (lane & 15)] = val;
                    }
            }
        }
    }
}

__global__ void __launch_bounds__(NT, 2)
jacobi_round_kernel(const JMat* __restrict__ mats, const JWork* __restrict__ work, int round, int max_inner,
                    unsigned long long* __restrict__ offmax_bits)
{
    // LDS: Gram | Qm | staging (Gram tiles: 2 x 32 x XS; wave partials: 4 x 32 x GS; update chunk: 32 x CS)
    constexpr int STAGE = 2 * JP * XS > 4 * JP * GS ? 2 * JP * XS : 4 * JP * GS;
    __shared__ __attribute__((aligned(16))) double smem[2 * JP * GS + JP * QS + STAGE + 2 * NPAIR + 8];
    __shared__ unsigned char pair_tab[(JP - 1) * NPAIR * 2];
    __shared__ int perm[JP];   // output row -> eigenvector column (descending eigenvalue)
    __shared__ int zrow[JP];   // 1: this row of the pair is numerically null (deflated)
    __shared__ int s_any_null;
    double* Gs = smem;
    double* G2 = Gs + JP * GS;
    double* Qs = G2 + JP * GS;
    double* Xc = Qs + JP * QS;
    double* cs = Xc + STAGE;
    double* red = cs + 2 * NPAIR;
    static_assert(STAGE >= JP * CS, "update chunk must fit in the staging area");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    const JWork wk = work[blockIdx.x];
    const JMat mt = mats[wk.mat];
    if (round >= mt.nb - 1) return; // this matrix has fewer rounds per sweep
    int P, Q;
    circle_pair(mt.nb, round, wk.slot, P, Q);
    if (P > Q) {
        const int t = P;
        P = Q;
        Q = t;
    }
    // round-robin schedule of the inner Jacobi (31 rounds x 16 pairs)
    for (int e = tid; e < (JP - 1) * NPAIR; e += NT) {
        int i, j;
        circle_pair(JP, e / NPAIR, e % NPAIR, i, j);
        pair_tab[2 * e] = (unsigned char)i;
        pair_tab[2 * e + 1] = (unsigned char)j;
    }

    // ---- 1. Gram matrix G = X X^T: every wave takes a quarter of each 64-deep k tile ----------
    {
        gp W = (gp)mt.W;
        const int ld = mt.lenp;
        d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        // tile: 32 rows x 64 k = 1024 d2 / 256 threads = 4 each; v -> (row = v>>5, kv = v&31);
        // PD tiles in flight in registers, LDS double buffered, one barrier per tile.
        d2 reg[PD][4];
        gcp2 src[4];
        const int kvo = 2 * (tid & 31);
#pragma unroll
        for (int p = 0; p < 4; ++p) src[p] = (gcp2)(W + (int64_t)xrow((tid + p * NT) >> 5, P, Q) * ld + kvo);
        const int ntile = ld / GK;
#pragma unroll
        for (int d = 0; d < PD; ++d)
            if (d < ntile) {
#pragma unroll
                for (int p = 0; p < 4; ++p) reg[d][p] = src[p][d * (GK / 2)];
            }
        for (int t0 = 0; t0 < ntile; t0 += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int t = t0 + d;
                if (t < ntile) {
                    double* Xn = Xc + (t & 1) * (JP * XS);
#pragma unroll
                    for (int p = 0; p < 4; ++p) *reinterpret_cast<d2*>(Xn + ((tid + p * NT) >> 5) * XS + kvo) = reg[d][p];
                    __syncthreads();
                    if (t + PD < ntile) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) reg[d][p] = src[p][(t + PD) * (GK / 2)];
                    }
                    const double* ap = Xn + wave * 16 + (lane >> 4) + (lane & 15) * XS;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double x0 = ap[kk * 4], x1 = ap[kk * 4 + 16 * XS];
                        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x0, acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, acc[1][1], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads(); // all waves are done reading the staging tiles
        double* part = Xc + wave * (JP * GS);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(i * 16 + (lane >> 4) + 4 * r) * GS + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    __syncthreads();
    // sum the four K-slices and symmetrise
    for (int e = tid; e < JP * JP; e += NT) {
        const int i = e / JP, j = e % JP;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += Xc[w * (JP * GS) + i * GS + j] + Xc[w * (JP * GS) + j * GS + i];
        Gs[i * GS + j] = 0.5 * s;
    }
    __syncthreads();

    // ---- 2a. deflation: rows whose squared norm fell below the numerical-rank threshold are
    //          removed from the problem (zeroed); they are completed after convergence.
    if (tid == 0) s_any_null = 0;
    __syncthreads();
    if (tid < JP) {
        const double thr2 = mt.thr2 ? *((const GLOBAL_AS double*)mt.thr2) : 0.0;
        const double g = Gs[tid * GS + tid];
        const int z = (g > 0.0 && g <= thr2) ? 1 : 0;
        zrow[tid] = z;
        perm[tid] = tid;
        if (z) s_any_null = 1;
    }
    __syncthreads();
    const bool any_null = s_any_null != 0;
    if (any_null) {
        for (int e = tid; e < JP * JP; e += NT) {
            const int i = e / JP, j = e % JP;
            if (zrow[i] || zrow[j]) Gs[i * GS + j] = 0.0;
        }
        __syncthreads();
    }
    // ---- 2b. convergence measure; nothing to do if this pair is already orthogonal --------
    double off = gram_offmax(Gs, red, tid);
    if (tid == 0) atomicMax(offmax_bits + wk.mat, (unsigned long long)__double_as_longlong(off));
    if (off <= mt.tol && !any_null) return;

    // ---- 3. two-sided Jacobi eigh of G in LDS, Qm accumulated ----------------------------
    for (int e = tid; e < JP * QS; e += NT) Qs[e] = ((e / QS) == (e % QS)) ? 1.0 : 0.0;
    __syncthreads();
    // One barrier per round: G is double buffered (read Ga, write Gb), and every thread derives the
    // two rotations it needs (row pair pr, column pair pc) itself from Ga instead of waiting for 16
    // lanes to publish them.  The Qm entries a thread updates belong to column pair pc as well.
    double* Ga = Gs;
    double* Gb = G2;
    for (int sweep = 0; sweep < (off <= mt.tol ? 0 : max_inner); ++sweep) {
        for (int r = 0; r < JP - 1; ++r) {
            const unsigned char* tab = pair_tab + r * NPAIR * 2;
            const int pr = tid >> 4, pc = tid & 15;
            const int i = tab[2 * pr], j = tab[2 * pr + 1], k = tab[2 * pc], l = tab[2 * pc + 1];
            double c1, s1, c2, s2;
            jacobi_rot(Ga[i * GS + i], Ga[j * GS + j], Ga[i * GS + j], c1, s1);
            jacobi_rot(Ga[k * GS + k], Ga[l * GS + l], Ga[k * GS + l], c2, s2);
            {   // G <- R^T G R on one 2x2 sub-block per thread: rows (i,j), cols (k,l)
                const double gik = Ga[i * GS + k], gil = Ga[i * GS + l];
                const double gjk = Ga[j * GS + k], gjl = Ga[j * GS + l];
                const double hik = c1 * gik - s1 * gjk, hil = c1 * gil - s1 * gjl;
                const double hjk = s1 * gik + c1 * gjk, hjl = s1 * gil + c1 * gjl;
                double nik = c2 * hik - s2 * hil, nil = s2 * hik + c2 * hil;
                double njk = c2 * hjk - s2 * hjl, njl = s2 * hjk + c2 * hjl;
                if (pr == pc) { // the rotated 2x2 diagonal block is diagonal by construction
                    nil = 0.0;
                    njk = 0.0;
                }
                Gb[i * GS + k] = nik;
                Gb[i * GS + l] = nil;
                Gb[j * GS + k] = njk;
                Gb[j * GS + l] = njl;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) { // Qm <- Qm R : columns (k,l) of rows (tid>>4) and (tid>>4)+16
                const int row = (tid >> 4) + 16 * t;
                const double qk = Qs[row * QS + k], ql = Qs[row * QS + l];
                Qs[row * QS + k] = c2 * qk - s2 * ql;
                Qs[row * QS + l] = s2 * qk + c2 * ql;
            }
            __syncthreads();
            double* tsw = Ga;
            Ga = Gb;
            Gb = tsw;
        }
        const double off_in = gram_offmax(Ga, red, tid);
        if (off_in <= 0.25 * mt.tol) break;
    }
    __syncthreads();
    if (Ga != Gs) { // keep the final Gram (its diagonal orders the rows below) in Gs
        for (int e = tid; e < JP * GS; e += NT) Gs[e] = Ga[e];
        __syncthreads();
    }
    // de Rijk-style ordering inside the pair: larger norms to the lower rows (fewer sweeps)
    if (off > mt.tol && tid < JP) {
        const double g = Gs[tid * GS + tid];
        int rk = 0;
        for (int j = 0; j < JP; ++j) {
            const double gj = Gs[j * GS + j];
            rk += (gj > g || (gj == g && j < tid)) ? 1 : 0;
        }
        perm[rk] = tid;
    }
    __syncthreads();

    // ---- 4. X <- Qm^T X  and  J_PQ <- Qm^T J_PQ -------------------------------------------
    apply_update(mt.W, mt.lenp, mt.lenp, P, Q, Qs, Xc, tid, perm, zrow, true);
    if (mt.J) {
        __syncthreads();
        apply_update(mt.J, mt.nvp, mt.nvp, P, Q, Qs, Xc, tid, perm, zrow, false);
    }
}

} // namespace

int jacobi_orthogonalise(cyb_ctx_t ctx, const std::vector<JMat>& h_mats, int max_sweeps,
                         std::vector<int32_t>& sweeps_out)
{
    const int n = (int)h_mats.size();
    sweeps_out.assign((size_t)n, 0);
    if (n == 0) return CYB_OK;
    hipStream_t st = ctx->stream;

    std::vector<int> active;
    for (int i = 0; i < n; ++i) {
        if (h_mats[(size_t)i].nv <= 1) sweeps_out[(size_t)i] = 0; // nothing to orthogonalise
        else active.push_back(i);
    }
    unsigned long long* d_off = nullptr;
    CYB_HIP(hipMalloc(&d_off, sizeof(unsigned long long) * (size_t)n));
    std::vector<unsigned long long> h_off((size_t)n);
    std::vector<double> prev_off((size_t)n, 1e300);
    int status = CYB_OK;
    for (int sweep = 1; sweep <= max_sweeps && !active.empty(); ++sweep) {
        // work list: matrices with more blocks first, so that late rounds use a prefix of the grid
        std::vector<int> order = active;
        std::stable_sort(order.begin(), order.end(),
                         [&](int a, int b) { return h_mats[(size_t)a].nb > h_mats[(size_t)b].nb; });
        std::vector<JWork> wl;
        int max_nb = 0;
        for (int m : order) {
            const int nb = h_mats[(size_t)m].nb;
            max_nb = std::max(max_nb, nb);
            for (int k = 0; k < nb / 2; ++k) wl.push_back(JWork{m, k});
        }
        // descriptors are re-uploaded every sweep: a ring slot only lives for a few uploads
        void* d_mats_v = nullptr;
        status = ctx->upload(h_mats.data(), sizeof(JMat) * (size_t)n, &d_mats_v);
        if (status != CYB_OK) break;
        const JMat* d_mats = static_cast<const JMat*>(d_mats_v);
        void* d_wl = nullptr;
        status = ctx->upload(wl.data(), sizeof(JWork) * wl.size(), &d_wl);
        if (status != CYB_OK) break;
        if (hipMemsetAsync(d_off, 0, sizeof(unsigned long long) * (size_t)n, st) != hipSuccess) {
            status = CYB_ERR_HIP;
            break;
        }
        // inner sweeps: few while far from convergence (the outer iteration repeats anyway)
        // inner Jacobi sweeps per pair visit: measurements and a numpy model agree that more than two
        // buy no outer sweeps, and one is fastest overall (measured: 81 vs 88 ms on the chi=4096 list)
        static const int max_inner = getenv("CYB_JACOBI_INNER") ? atoi(getenv("CYB_JACOBI_INNER")) : 1;
        for (int r = 0; r < max_nb - 1; ++r) {
            // grid = prefix of the work list holding matrices with nb - 1 > r
            size_t cnt = 0;
            for (int m : order) {
                if (h_mats[(size_t)m].nb - 1 > r) cnt += (size_t)h_mats[(size_t)m].nb / 2;
                else break;
            }
            if (cnt == 0) break;
            hipLaunchKernelGGL(jacobi_round_kernel, dim3((unsigned)cnt), dim3(NT), 0, st, d_mats,
                               static_cast<const JWork*>(d_wl), r, max_inner, d_off);
        }
        if (hipGetLastError() != hipSuccess) {
            set_error("jacobi_round_kernel launch failed");
            status = CYB_ERR_HIP;
            break;
        }
        if (hipMemcpyAsync(h_off.data(), d_off, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost, st) !=
                hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            set_error("jacobi: reading the convergence flags failed: %s", hipGetErrorString(hipGetLastError()));
            status = CYB_ERR_HIP;
            break;
        }
        std::vector<int> still;
        static const bool trace = getenv("CYB_JACOBI_TRACE") != nullptr;
        for (int m : active) {
            double off;
            memcpy(&off, &h_off[(size_t)m], sizeof(double));
            if (trace)
                fprintf(stderr, "[jacobi] sweep %2d mat %3d (nv=%d len=%d) off=%.3e tol=%.3e\n", sweep, m,
                        h_mats[(size_t)m].nv, h_mats[(size_t)m].len, off, h_mats[(size_t)m].tol);
            const double tol = h_mats[(size_t)m].tol;
            // converged, or stagnated within a small factor of the threshold (rounding floor of the
            // Gram products for long vectors)
            const bool stagnated = sweep >= 6 && off <= 64.0 * tol && off >= 0.5 * prev_off[(size_t)m];
            if (off <= tol || stagnated) sweeps_out[(size_t)m] = sweep;
            else still.push_back(m);
            prev_off[(size_t)m] = off;
        }
        active.swap(still);
    }
    (void)hipFree(d_off);
    if (status != CYB_OK) return status;
    if (!active.empty()) {
        for (int m : active) sweeps_out[(size_t)m] = -1;
        set_error("block-Jacobi did not converge within %d sweeps for %zu matrices", max_sweeps, active.size());
        return CYB_ERR_NOCONV;
    }
    return CYB_OK;
}

} // namespace cyb
