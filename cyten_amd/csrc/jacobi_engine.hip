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

namespace cyb {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define GLOBAL_AS __attribute__((address_space(1)))
typedef const GLOBAL_AS d2* gc2;
typedef GLOBAL_AS double* gp;

constexpr int GS = 66;  // row stride of the Gram matrix in LDS
constexpr int QS = 80;  // row stride of Qm / update chunk in LDS ([k][m] layout: 80 = 16 mod 32)
constexpr int XS = 18;  // row stride of the Gram staging tile ([m][k] layout, 16 + 2)
constexpr int NT = 256;

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

// scaled off-diagonal measure of the 64x64 Gram matrix in LDS
__device__ __forceinline__ double gram_offmax(const double* Gs, double* red, int tid)
{
    double m = 0.0;
#pragma unroll 4
    for (int e = tid; e < JP * JP; e += NT) {
        const int i = e >> 6, j = e & 63;
        if (i < j) {
            const double gii = Gs[i * GS + i], gjj = Gs[j * GS + j];
            const double den = gii * gjj;
            if (den > 0.0) m = fmax(m, fabs(Gs[i * GS + j]) * rsqrt(den));
        }
    }
    return block_max(m, red, tid);
}

// pair of players meeting in round r (0..n-2), slot k (0..n/2-1) of the circle method, n even
__device__ __forceinline__ void circle_pair(int n, int r, int k, int& p, int& q)
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

__device__ __forceinline__ int xrow(int i, int P, int Q) { return (i < JB) ? P * JB + i : Q * JB + (i - JB); }

// out(64 x ncols) = Qm^T X for the 64 rows {P-block, Q-block} of the row-major matrix `base`
// (row stride ld, ncols a multiple of 64), in place.
__device__ __forceinline__ void apply_update(double* __restrict__ base_, int ld, int ncols, int P, int Q,
                                             const double* Qs, double* Xc, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    gp base = (gp)base_;
    d2 reg[8];
    // prefetch chunk 0: 64 rows x 64 cols = 2048 d2 / 256 threads = 8 each; v -> (row = v>>5, cv = v&31)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int v = tid + p * NT;
        reg[p] = *(gc2)(base + (int64_t)xrow(v >> 5, P, Q) * ld + 2 * (v & 31));
    }
    for (int c0 = 0; c0 < ncols; c0 += 64) {
        __syncthreads(); // previous chunk's LDS reads are done
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int v = tid + p * NT;
            *reinterpret_cast<d2*>(Xc + (v >> 5) * QS + 2 * (v & 31)) = reg[p];
        }
        __syncthreads();
        if (c0 + 64 < ncols) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int v = tid + p * NT;
                reg[p] = *(gc2)(base + (int64_t)xrow(v >> 5, P, Q) * ld + c0 + 64 + 2 * (v & 31));
            }
        }
        d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        // A[m][k] = Qm[k][m]  (Qs is [k][m]);  B[k][n] = Xc[k][n]
        const double* ap = Qs + (lane >> 4) * QS + wm * 32 + (lane & 15);
        const double* bp = Xc + (lane >> 4) * QS + wn * 32 + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < JP / 4; ++kk) {
            double a[2], b[2];
            a[0] = ap[kk * 4 * QS];
            a[1] = ap[kk * 4 * QS + 16];
            b[0] = bp[kk * 4 * QS];
            b[1] = bp[kk * 4 * QS + 16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 32 + i * 16 + (lane >> 4) + 4 * r;
                gp orow = base + (int64_t)xrow(row, P, Q) * ld + c0 + wn * 32 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 2; ++j) orow[j * 16] = acc[i][j][r];
            }
    }
}

__global__ void __launch_bounds__(NT, 1)
jacobi_round_kernel(const JMat* __restrict__ mats, const JWork* __restrict__ work, int round, int max_inner,
                    unsigned long long* __restrict__ offmax_bits)
{
    __shared__ __attribute__((aligned(16))) double smem[JP * GS + JP * QS + JP * QS + 64 + 8];
    double* Gs = smem;
    double* Qs = Gs + JP * GS;
    double* Xc = Qs + JP * QS; // Gram staging (2 x 64 x XS) and update chunk (64 x QS)
    double* cs = Xc + JP * QS;
    double* red = cs + 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

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

    // ---- 1. Gram matrix G = X X^T --------------------------------------------------------
    {
        gp W = (gp)mt.W;
        const int ld = mt.lenp;
        d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        d2 reg[2];
        // tile: 64 rows x 16 k = 512 d2 / 256 threads = 2 each; v -> (row = v>>3, kv = v&7)
        const int r0 = xrow(tid >> 3, P, Q), r1 = xrow((tid + NT) >> 3, P, Q);
        const int kvo = 2 * (tid & 7);
        reg[0] = *(gc2)(W + (int64_t)r0 * ld + kvo);
        reg[1] = *(gc2)(W + (int64_t)r1 * ld + kvo);
        *reinterpret_cast<d2*>(Xc + (tid >> 3) * XS + kvo) = reg[0];
        *reinterpret_cast<d2*>(Xc + ((tid + NT) >> 3) * XS + kvo) = reg[1];
        __syncthreads();
        int buf = 0;
        for (int k0 = 0; k0 < ld; k0 += 16) {
            const bool have_next = (k0 + 16 < ld);
            if (have_next) {
                reg[0] = *(gc2)(W + (int64_t)r0 * ld + k0 + 16 + kvo);
                reg[1] = *(gc2)(W + (int64_t)r1 * ld + k0 + 16 + kvo);
            }
            const double* Xs = Xc + buf * (JP * XS);
            const double* ap = Xs + (wm * 32 + (lane & 15)) * XS + (lane >> 4);
            const double* bp = Xs + (wn * 32 + (lane & 15)) * XS + (lane >> 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double a[2], b[2];
                a[0] = ap[kk * 4];
                a[1] = ap[kk * 4 + 16 * XS];
                b[0] = bp[kk * 4];
                b[1] = bp[kk * 4 + 16 * XS];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (have_next) {
                buf ^= 1;
                double* Xn = Xc + buf * (JP * XS);
                *reinterpret_cast<d2*>(Xn + (tid >> 3) * XS + kvo) = reg[0];
                *reinterpret_cast<d2*>(Xn + ((tid + NT) >> 3) * XS + kvo) = reg[1];
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Gs[(wm * 32 + i * 16 + (lane >> 4) + 4 * r) * GS + wn * 32 + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    __syncthreads();
    // symmetrise (the two triangles come from different accumulation orders)
    for (int e = tid; e < JP * JP; e += NT) {
        const int i = e >> 6, j = e & 63;
        if (i < j) {
            const double s = 0.5 * (Gs[i * GS + j] + Gs[j * GS + i]);
            Gs[i * GS + j] = s;
            Gs[j * GS + i] = s;
        }
    }
    __syncthreads();

    // ---- 2. convergence measure; nothing to do if this pair is already orthogonal ---------
    double off = gram_offmax(Gs, red, tid);
    if (tid == 0) atomicMax(offmax_bits + wk.mat, (unsigned long long)__double_as_longlong(off));
    if (off <= mt.tol) return;

    // ---- 3. two-sided Jacobi eigh of G in LDS, Qm accumulated ----------------------------
    for (int e = tid; e < JP * QS; e += NT) Qs[e] = 0.0;
    __syncthreads();
    if (tid < JP) Qs[tid * QS + tid] = 1.0;
    __syncthreads();
    for (int sweep = 0; sweep < max_inner; ++sweep) {
        for (int r = 0; r < JP - 1; ++r) {
            if (tid < JP / 2) {
                int i, j;
                circle_pair(JP, r, tid, i, j);
                const double a = Gs[i * GS + i], d = Gs[j * GS + j], b = Gs[i * GS + j];
                double c = 1.0, s = 0.0;
                if (b != 0.0 && fabs(b) > 1e-300) {
                    const double tau = (d - a) / (2.0 * b);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                }
                cs[2 * tid] = c;
                cs[2 * tid + 1] = s;
            }
            __syncthreads();
            // G <- R^T G R on 2x2 sub-blocks: rows (i,j) = pair pr, cols (k,l) = pair pc
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int bidx = tid + t * NT;
                const int pr = bidx >> 5, pc = bidx & 31;
                int i, j, k, l;
                circle_pair(JP, r, pr, i, j);
                circle_pair(JP, r, pc, k, l);
                const double c1 = cs[2 * pr], s1 = cs[2 * pr + 1];
                const double c2 = cs[2 * pc], s2 = cs[2 * pc + 1];
                const double gik = Gs[i * GS + k], gil = Gs[i * GS + l];
                const double gjk = Gs[j * GS + k], gjl = Gs[j * GS + l];
                // rows: row_i' = c1 row_i - s1 row_j ; row_j' = s1 row_i + c1 row_j
                const double hik = c1 * gik - s1 * gjk, hil = c1 * gil - s1 * gjl;
                const double hjk = s1 * gik + c1 * gjk, hjl = s1 * gil + c1 * gjl;
                // cols: col_k' = c2 col_k - s2 col_l ; col_l' = s2 col_k + c2 col_l
                double nik = c2 * hik - s2 * hil, nil = s2 * hik + c2 * hil;
                double njk = c2 * hjk - s2 * hjl, njl = s2 * hjk + c2 * hjl;
                if (pr == pc) { // the rotated 2x2 diagonal block is diagonal by construction
                    nil = 0.0;
                    njk = 0.0;
                }
                Gs[i * GS + k] = nik;
                Gs[i * GS + l] = nil;
                Gs[j * GS + k] = njk;
                Gs[j * GS + l] = njl;
            }
            // Qm <- Qm R : columns (i,j) of every row
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int e = tid + t * NT;
                const int pr = e & 31, row = e >> 5;
                int i, j;
                circle_pair(JP, r, pr, i, j);
                const double c1 = cs[2 * pr], s1 = cs[2 * pr + 1];
                const double qi = Qs[row * QS + i], qj = Qs[row * QS + j];
                Qs[row * QS + i] = c1 * qi - s1 * qj;
                Qs[row * QS + j] = s1 * qi + c1 * qj;
            }
            __syncthreads();
        }
        const double off_in = gram_offmax(Gs, red, tid);
        if (off_in <= 0.25 * mt.tol) break;
    }
    __syncthreads();

    // ---- 4. X <- Qm^T X  and  J_PQ <- Qm^T J_PQ -------------------------------------------
    apply_update(mt.W, mt.lenp, mt.lenp, P, Q, Qs, Xc, tid);
    if (mt.J) {
        __syncthreads();
        apply_update(mt.J, mt.nvp, mt.nvp, P, Q, Qs, Xc, tid);
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
        const int max_inner = sweep <= 2 ? 2 : 4;
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
        for (int m : active) {
            double off;
            memcpy(&off, &h_off[(size_t)m], sizeof(double));
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
