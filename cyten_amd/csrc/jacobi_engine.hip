// One-sided block-Jacobi orthogonalisation on gfx950 (engine of the batched SVD / eigh).
//
// Replaces the per-block LAPACK calls of the reference (scipy.linalg.svd -> dgesdd,
// src/block_backend/numpy.cpp:1247-1297; np.linalg.eigh -> dsyevd, numpy.cpp:658-680) with a
// Hestenes one-sided Jacobi in *block* form so that the O(len * nv^2) work per sweep runs on the
// f64 MFMA pipe:
//   for every round of the round-robin schedule over blocks of JB=16 vectors, a block pair (P,Q) of one
//   matrix goes through TWO launches:
//     A. jacobi_gram_kernel (G workgroups per pair)
//          Gram   G = X X^T       X = [W_P; W_Q] (32 x len), MFMA 16x16x4 f64 straight from global memory,
//                                 K = len split over the G parts, last arriver sums the partials
//          eigh   G ~ Qm L Qm^T   ONE sweep of two-sided Jacobi on the 32x32 Gram matrix in LDS (16 disjoint
//                                 rotations per inner round, one barrier each), deflation of numerically null
//                                 rows, in-pair descending sort, Qm published
//          (+ workgroups that apply the PREVIOUS round's Qm to J, which nothing reads before the end)
//     B. jacobi_update_kernel     X <- Qm^T X on W (64-column chunks, MFMA, operands straight from global)
//   Rounds are separate launches (the next round needs this round's rows); the per-sweep
//   convergence measure max |g_ij|/sqrt(g_ii g_jj) is accumulated with an atomic max and read by
//   the host once per sweep.
#include "jacobi_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace cyb {
constexpr double kPredictQuad = 16.0; // the last sweep is predicted only behind a step with off <= kPredictQuad * prev^2 (jacobi_orthogonalise)
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define GLOBAL_AS __attribute__((address_space(1)))
typedef const GLOBAL_AS d2* gc2;
typedef const GLOBAL_AS d2* gcp2;
typedef GLOBAL_AS double* gp;

constexpr int GS = JP + 2;   // row stride of the Gram matrix in LDS
constexpr int QS = JP + 16;  // row stride of Qm in LDS ([k][m] layout: 48 = 16 mod 32 -> conflict-free)
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

// v_rsq_f64 refined with two Newton steps (full double precision for the rotation)
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}

// Symmetric 2x2 Jacobi rotation [[c, s], [-s, c]] that annihilates b in [[a, b], [b, d]] (the small
// angle |theta| <= pi/4).  Division-free: with delta = d - a, h = hypot(delta, 2b):
//   cos 2theta = |delta| / h,  c = sqrt((1 + cos 2theta) / 2),  s = sign(delta b) |b| / (h c).
// Two v_rsq_f64 + a dozen FMAs instead of two IEEE divisions and two square roots: the inner
// eigensolver runs one wave per SIMD, so every dependent long-latency op is exposed.
__device__ __forceinline__ void jacobi_rot(double a, double d, double b, double& c, double& s)
{
    const double delta = d - a;
    const double h2 = delta * delta + 4.0 * b * b;
    c = 1.0;
    s = 0.0;
    if (fabs(b) > 1e-300 && h2 > 1e-300) {
        const double rh = fast_rsqrt(h2);            // 1 / h
        const double c2 = 0.5 + 0.5 * fabs(delta) * rh; // cos^2 theta in [1/2, 1]
        const double rc = fast_rsqrt(c2);            // 1 / c
        c = c2 * rc;
        const double sg = ((delta >= 0.0) == (b >= 0.0)) ? 1.0 : -1.0;
        s = sg * fabs(b) * rh * rc;
    }
}

__device__ __forceinline__ int xrow(int i, int P, int Q) { return (i < JB) ? P * JB + i : Q * JB + (i - JB); }

// Per-round scratch shared by the two kernels of a round (one entry per block pair of the round).
struct JScratch {
    double* gpart;        // [pair][part][JP*JP]  partial Gram matrices
    // the three result arrays of a round exist twice (index = round parity): the J half of a round's
    // update is applied one launch later, beside the next round's Gram/eigensolve
    double* qout[2];      // [pair][JP*JP]        Qm as [k][m], output-row order
    int32_t* zout[2];     // [pair][JP]           1: output row is deflated (write zeros)
    int32_t* flag[2];     // [pair]               1: apply the update, 0: pair already orthogonal
    unsigned int* cnt;    // [pair]               arrival counter of the Gram parts (self-resetting)
};

// Kernel A of a round.  grid = pairs x G.  Every workgroup computes the Gram partial of its K range;
// the last of the G parts of a pair to arrive (write-through stores, drained, one atomic ticket,
// L1-bypassing loads: cdna_hip_programming.md Guideline 16) sums the partials in a FIXED order, then does the
// deflation test, the convergence measure and the two-sided Jacobi eigh of the 32x32 Gram in LDS,
// and publishes Qm for kernel B.  No workgroup ever waits for another one.
__device__ __forceinline__ void update_role(const JMat* __restrict__ mats, const JWork* __restrict__ work, int round,
                                            int pi, int unit, int U, const JScratch& sc, int buf, bool do_w, bool do_j,
                                            double* smem, int* zflag);

// Workgroups beyond the first n_gram of the grid take the update role for the J half of the PREVIOUS
// round (prev_round, hand-off buffer buf ^ 1): nothing reads J before the end, so that half of the
// update leaves the critical path and runs beside this round's Gram / eigensolve.
__global__ void __launch_bounds__(NT, 2)
jacobi_gram_kernel(const JMat* __restrict__ mats, const JWork* __restrict__ work, int round, int G, int max_inner,
                   unsigned long long* __restrict__ offmax_bits, JScratch sc, int buf, int n_gram, int prev_round, int UJ)
{
    constexpr int STAGE = 2 * JP * XS > 4 * JP * GS ? 2 * JP * XS : 4 * JP * GS;
    __shared__ __attribute__((aligned(16))) double smem[2 * JP * GS + JP * QS + STAGE + 8];
    __shared__ int perm[JP];   // output row -> eigenvector column (descending eigenvalue)
    __shared__ int zrow[JP];   // 1: this row of the pair is numerically null (deflated)
    __shared__ int s_any_null, s_last;
    __shared__ unsigned short sched[(JP - 1) * NPAIR];
    double* Gs = smem;
    double* G2 = Gs + JP * GS;
    double* Qs = G2 + JP * GS;
    double* Xc = Qs + JP * QS;
    double* red = Xc + STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= n_gram) {
        const int b = (int)blockIdx.x - n_gram;
        update_role(mats, work, prev_round, b / UJ, b % UJ, UJ, sc, buf ^ 1, false, true, smem, zrow);
        return;
    }
    const int pi = blockIdx.x / G, part = blockIdx.x % G;

    const JWork wk = work[pi];
    const JMat mt = mats[wk.mat];
    if (round >= mt.nb - 1) return; // this matrix has fewer rounds per sweep
    int P, Q;
    circle_pair(mt.nb, round, wk.slot, P, Q);
    if (P > Q) {
        const int t = P;
        P = Q;
        Q = t;
    }

    // ---- 1. partial Gram over this part's share of K
    {
        gp W = (gp)mt.W;
        const int ld = mt.lenp;
        d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        // Operands go from global memory straight into the MFMA: lane (r = lane & 15, g = lane >> 4)
        // loads the two doubles X[r][k0 + 2g], X[r][k0 + 2g + 1] of rows r and r + 16 (16-byte loads,
        // 64 contiguous bytes per row and wave) and feeds .x to one MFMA k-step and .y to the next --
        // both operands of a Gram product are the same matrix, so any pairing of k positions that is
        // the same for the two of them is right.  No staging tile, no barrier in the loop; the four
        // waves take the 8-deep k chunks round-robin.
        const int r = lane & 15, g2 = 2 * (lane >> 4);
        gcp2 s0 = (gcp2)(W + (int64_t)xrow(r, P, Q) * ld + g2);
        gcp2 s1 = (gcp2)(W + (int64_t)xrow(r + 16, P, Q) * ld + g2);
        const int nchunk_all = ld / 8; // ld is a multiple of 64 (zero padded)
        const int c_begin = (int)((int64_t)part * nchunk_all / G), c_end = (int)((int64_t)(part + 1) * nchunk_all / G);
        // the loop is bound by load latency (the rows were written by the previous launch): UN chunks
        // per wave are in flight while the previous UN are multiplied
        constexpr int UN = 8;
        d2 x0[UN], x1[UN], n0[UN], n1[UN];
        auto load = [&](d2 (&a0)[UN], d2 (&a1)[UN], int c0) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int c = min(c0 + 4 * u, c_end - 1); // clamped: a duplicate is masked out in the product
                a0[u] = s0[c * 4];
                a1[u] = s1[c * 4];
            }
        };
        if (c_begin + wave < c_end) load(x0, x1, c_begin + wave);
        for (int c0 = c_begin + wave; c0 < c_end; c0 += 4 * UN) {
            const bool more = c0 + 4 * UN < c_end; // wave-uniform
            if (more) load(n0, n1, c0 + 4 * UN);
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (c0 + 4 * u < c_end) { // wave-uniform
                    // the (1,0) tile is the transpose of (0,1): not computed (the phase is bound by
                    // the MFMA pipe of the one or two CUs that work on a pair)
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].x, x0[u].x, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].x, x1[u].x, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u].x, x1[u].x, acc[1][1], 0, 0, 0);
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].y, x0[u].y, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].y, x1[u].y, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u].y, x1[u].y, acc[1][1], 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    x0[u] = n0[u];
                    x1[u] = n1[u];
                }
            }
        }
        double* wpart = Xc + wave * (JP * GS);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (i == 1 && j == 0) // mirror of tile (0,1)
                        wpart[(16 + (lane & 15)) * GS + (lane >> 4) + 4 * r] = acc[0][1][r];
                    else
                        wpart[(i * 16 + (lane >> 4) + 4 * r) * GS + j * 16 + (lane & 15)] = acc[i][j][r];
                }
    }
    __syncthreads();
    if (G == 1) {
        // sum the four wave slices and symmetrise, straight into Gs
        for (int e = tid; e < JP * JP; e += NT) {
            const int i = e / JP, j = e % JP;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += Xc[w * (JP * GS) + i * GS + j];
            Gs[i * GS + j] = s;
        }
        __syncthreads();
    } else {
        // publish this part's Gram partial (sum over the wave slices), then take a ticket
        GLOBAL_AS double* mine = (GLOBAL_AS double*)(sc.gpart + ((size_t)pi * G + part) * (JP * JP));
        for (int e = tid; e < JP * JP; e += NT) {
            const int i = e / JP, j = e % JP;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += Xc[w * (JP * GS) + i * GS + j];
            // EVERY store of the handed-off bytes is an 8-byte agent-scope (write-through, sc1) store ...
            __hip_atomic_store((double*)(mine + e), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // ... drained by every storing wave ...
        __syncthreads();                                  // ... before ONE lane takes the ticket
        if (tid == 0) {
            const unsigned int old = __hip_atomic_fetch_add(sc.cnt + pi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == (unsigned int)(G - 1)) ? 1 : 0;
            if (s_last) __hip_atomic_store(sc.cnt + pi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-arm
        }
        __syncthreads();
        if (!s_last) return;
        // the last arriver reads every partial with agent-scope (sc1, L1-bypassing) loads: no fence needed
        // (MI355X_MICROARCH.md, valid hand-off forms: sc1 stores + drained counter add + sc1 loads)
        const GLOBAL_AS double* all = (const GLOBAL_AS double*)(sc.gpart + (size_t)pi * G * (JP * JP));
        for (int e = tid; e < JP * JP; e += NT) {
            const int i = e / JP, j = e % JP;
            double s = 0.0;
            // (every partial is exactly symmetric: the diagonal tiles are X X^T products accumulated in
            //  the same order on both sides of the diagonal, the (1,0) tile is a copy of (0,1))
            for (int g = 0; g < G; ++g)
                s += __hip_atomic_load((double*)(all + (size_t)g * (JP * JP) + i * JP + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            Gs[i * GS + j] = s;
        }
        __syncthreads();
    }
    // ---- 2a. deflation: rows whose squared norm fell below the numerical-rank threshold are
    //          removed from the problem (zeroed); they are completed after convergence.
    if (tid == 0) s_any_null = 0;
    __syncthreads();
    if (tid < JP) {
        const double thr2 = mt.thr2;
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
    if (tid == 0) {
        atomicMax(offmax_bits + wk.mat, (unsigned long long)__double_as_longlong(off));
        sc.flag[buf][pi] = (off <= mt.tol && !any_null) ? 0 : 1;
    }
    if (off <= mt.tol && !any_null) return;

    // ---- 3. two-sided Jacobi eigh of G in LDS, Qm accumulated ----------------------------
    for (int e = tid; e < JP * QS; e += NT) Qs[e] = ((e / QS) == (e % QS)) ? 1.0 : 0.0;
    for (int e = tid; e < (JP - 1) * NPAIR; e += NT) { // round-robin schedule (circle method) as a table
        int p, q;
        circle_pair(JP, e / NPAIR, e % NPAIR, p, q);
        sched[e] = (unsigned short)(p | (q << 8));
    }
    __syncthreads();
    // One barrier per round: G is double buffered (read Ga, write Gb).  The eigensolver runs one wave
    // per SIMD, so it is bound by the instruction count per round: every thread derives ONE rotation
    // itself (its column pair pc; 16 lanes compute the same one instead of waiting for a publish
    // through LDS) and takes the rotation of its row pair pr from the lane of its 16-lane group that
    // has pc == pr; the pair indices come from the table.
    double* Ga = Gs;
    double* Gb = G2;
    for (int sweep = 0; sweep < (off <= mt.tol ? 0 : max_inner); ++sweep) {
        for (int r = 0; r < JP - 1; ++r) {
            const int pr = tid >> 4, pc = tid & 15;
            const unsigned int sr = sched[r * NPAIR + pr], scl = sched[r * NPAIR + pc];
            const int i = sr & 255, j = sr >> 8, k = scl & 255, l = scl >> 8;
            double c2, s2;
            jacobi_rot(Ga[k * GS + k], Ga[l * GS + l], Ga[k * GS + l], c2, s2);
            const int srcl = (lane & 48) | pr;
            const double c1 = __shfl(c2, srcl), s1 = __shfl(s2, srcl);
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
        if (sweep + 1 < max_inner) { // (no test after the last sweep: two barriers for nothing)
            const double off_in = gram_offmax(Ga, red, tid);
            if (off_in <= 0.25 * mt.tol) break;
        }
    }
    __syncthreads();
    // de Rijk-style ordering inside the pair: larger norms to the lower rows (fewer sweeps)
    if (off > mt.tol && tid < JP) {
        // padding vectors (index >= nv: zero rows of W, identity rows of J) must stay behind the real
        // ones whatever the rounding of the rotated diagonal says: a real row whose eigenvalue came out as
        // -1e-16 would otherwise trade places with a padding row and leave the first nv rows for good
        // (exactly rank-deficient inputs: an all-ones block lost two rows of its accumulated factor)
        auto key = [&](int i) { return xrow(i, P, Q) < mt.nv ? Ga[i * GS + i] : -1.0e300; };
        const double g = key(tid);
        int rk = 0;
        for (int j = 0; j < JP; ++j) {
            const double gj = key(j);
            rk += (gj > g || (gj == g && j < tid)) ? 1 : 0;
        }
        perm[rk] = tid;
    }
    __syncthreads();
    // ---- 4. publish Qm in output-row order (column perm[r] of Qm feeds output row r) -----
    GLOBAL_AS double* qo = (GLOBAL_AS double*)(sc.qout[buf] + (size_t)pi * (JP * JP));
    for (int e = tid; e < JP * JP; e += NT) {
        const int k = e / JP, r = e % JP;
        qo[e] = Qs[k * QS + perm[r]];
    }
    if (tid < JP) sc.zout[buf][(size_t)pi * JP + tid] = zrow[perm[tid]];
}

// Update role: unit `unit` of U of pair `pi` applies X <- Qm^T X to its share of the 64-column chunks
// of W (do_w) and / or J (do_j), with the round's result taken from hand-off buffer `buf`.
// `smem` needs JP*QS doubles, `zflag` JP ints.
// The rows go from global memory straight into the MFMA B operand (lane (n = lane & 15, kq = lane >> 4)
// loads X[4 kk + kq][16 wave + n], 128 contiguous bytes per row and load) and the result straight back:
// each wave owns 16 columns of a chunk, so nothing is staged in LDS, there is no barrier in the loop, and
// the first chunk's loads are in flight while Qm is still being fetched.
__device__ __forceinline__ void update_role(const JMat* __restrict__ mats, const JWork* __restrict__ work, int round,
                                            int pi, int unit, int U, const JScratch& sc, int buf, bool do_w, bool do_j,
                                            double* smem, int* zflag)
{
    double* Qs = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const JWork wk = work[pi];
    const JMat mt = mats[wk.mat];
    if (round >= mt.nb - 1) return;
    if (sc.flag[buf][pi] == 0) return; // pair already orthogonal
    int P, Q;
    circle_pair(mt.nb, round, wk.slot, P, Q);
    if (P > Q) {
        const int t = P;
        P = Q;
        Q = t;
    }
    const int cw = do_w ? mt.lenp / 64 : 0, cj = (do_j && mt.J) ? mt.nvp / 64 : 0;
    const int ct = cw + cj;
    const int c_begin = (int)((int64_t)unit * ct / U), c_end = (int)((int64_t)(unit + 1) * ct / U);
    if (c_begin >= c_end) return;
    // operand pointer of chunk c (the unit's range may straddle the W | J boundary)
    const int n = lane & 15, kq = lane >> 4;
    auto chunk_ptr = [&](int c, int k) -> gp {
        const bool in_w = c < cw;
        gp base = (gp)(in_w ? mt.W : mt.J);
        const int ld = in_w ? mt.lenp : mt.nvp;
        const int cc = in_w ? c : c - cw;
        return base + (int64_t)xrow(k, P, Q) * ld + cc * 64 + wave * 16 + n;
    };
    double xb[JP / 4], xn[JP / 4];
#pragma unroll
    for (int kk = 0; kk < JP / 4; ++kk) xb[kk] = *chunk_ptr(c_begin, 4 * kk + kq);
    const GLOBAL_AS double* qo = (const GLOBAL_AS double*)(sc.qout[buf] + (size_t)pi * (JP * JP));
    for (int e = tid; e < JP * JP; e += NT) Qs[(e / JP) * QS + (e % JP)] = qo[e];
    if (tid < JP) zflag[tid] = sc.zout[buf][(size_t)pi * JP + tid];
    __syncthreads();
    const double* ap0 = Qs + (lane >> 4) * QS + (lane & 15);
    const double* ap1 = ap0 + 16;
    int zr[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) zr[i][r] = zflag[i * 16 + (lane >> 4) + 4 * r];
    for (int c = c_begin; c < c_end; ++c) {
        if (c + 1 < c_end) {
#pragma unroll
            for (int kk = 0; kk < JP / 4; ++kk) xn[kk] = *chunk_ptr(c + 1, 4 * kk + kq);
        }
        d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
        // A[m][k] = Qm[k][m]  (Qs is [k][m]);  B[k][n] = X[k][n]
#pragma unroll
        for (int kk = 0; kk < JP / 4; ++kk) {
            const double a0 = ap0[kk * 4 * QS], a1 = ap1[kk * 4 * QS];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xb[kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, xb[kk], acc[1], 0, 0, 0);
        }
        const bool zero_null = c < cw; // deflated rows are zeroed in W only
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + (lane >> 4) + 4 * r;
                const double val = (zero_null && zr[i][r]) ? 0.0 : acc[i][r];
                *chunk_ptr(c, row) = val;
            }
#pragma unroll
        for (int kk = 0; kk < JP / 4; ++kk) xb[kk] = xn[kk];
    }
}

// Kernel B of a round.  grid = pairs x U.
__global__ void __launch_bounds__(NT, 3)
jacobi_update_kernel(const JMat* __restrict__ mats, const JWork* __restrict__ work, int round, int U, JScratch sc, int buf,
                     int do_w, int do_j)
{
    __shared__ __attribute__((aligned(16))) double smem[JP * QS];
    __shared__ int zflag[JP];
    update_role(mats, work, round, blockIdx.x / U, blockIdx.x % U, U, sc, buf, do_w != 0, do_j != 0, smem, zflag);
}


// ================================================================================================
// Fused round: ONE launch per round of the round-robin schedule (the default path).
//
// grid = pairs x G.  The G workgroups of a block pair own disjoint COLUMN shares of the pair's 32 rows
// (whole 64-column chunks of W, and of J) for the whole launch:
//   0. ONE 64-byte descriptor read (pair + matrix), then every global load of the launch is issued up front: the own W
//      share in Gram layout and -- for shares of up to six chunks -- the own W / J share in update layout, which waits
//      in registers behind the eigensolve
//   1. partial Gram over the own W share (MFMA, operands straight from memory)
//   2. G > 1: every part publishes its 32x32 partial (16-byte write-through stores, drained, one arrival ticket per
//      workgroup) and ALL G parts wait for the G tickets (one lane polls, bounded), then every part reads the G
//      partials (16-byte sc1 loads, all in flight at once) and sums them in the same fixed order -- bit-identical
//      Gram matrices in all of them
//   3. deflation test, convergence measure, two-sided Jacobi eigensolve of the 32x32 Gram, REDUNDANTLY in every part
//      (the chip is idle otherwise; no second exchange, no publish of Qm)
//   4. X <- Qm^T X on the own column share of W and J, in place
// No workgroup reads or writes a column another workgroup of the launch touches, so rows need no hand-off inside
// the launch; the only inter-workgroup data are the 8 KB partials (cdna_hip_programming.md Guideline 16: sc1
// stores, every storing wave drained, one agent-scope ticket per workgroup, one polling lane, workgroup barrier,
// sc1 loads).  The wait needs the G parts of a pair to be co-resident: the host keeps grid <= number of CUs and the
// kernel asks for more than half a CU's LDS, so every workgroup has a CU of its own.  Every spin is bounded (wall
// clock); a timeout raises a flag the host turns into an error.
//
// Eigensolver, second version: the Gram matrix and the accumulated rotations live in LDS in POSITION space --
// thread (pr, pc) always owns the 2x2 block at block position (pr, pc), the matrix is physically permuted by the
// writes of every step (Brent-Luk ring: position 0 fixed, the other 31 advance one place), so a step is ONE LDS
// round trip (own block + the two diagonal blocks that define the thread's row and column rotation, both derived
// redundantly by every thread, branch-free so that the two dependent chains interleave), the 2x2 updates, eight
// scattered stores and one barrier: no schedule table, no lane exchange.  After 31 steps every index is back at its
// position.
struct RPair {            // everything a workgroup needs to know about its pair: one 64-byte read
    double* W;
    double* J;            // may be null
    double tol, thr2;
    int32_t nvp, lenp, nb, nv;
    int32_t mat, slot;
    int32_t pad[2];
};
static_assert(sizeof(RPair) == 64, "RPair is read as one 64-byte descriptor");

struct RScratch {
    double* gpart;       // [pair][G][JP*JP]  partial Gram matrices
    unsigned int* cnt;   // [round][pair]     arrival tickets (zeroed once per sweep)
    unsigned int* err;   // [1]               set when a bounded wait ran out
    int np;              // pairs per round row of cnt
    unsigned long long* stamps; // diagnostic runs only (CYB_JACOBI_STAMPS): [workgroup][8] wall-clock stamps, else null
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
union Q16 {               // one 16-byte memory word = two doubles
    u32x4 u;
    d2 d;
};

constexpr int VS = JP + 2;                                      // row stride of V in LDS (position space)
constexpr int ROUND_LDS_DOUBLES = 2 * JP * GS + 2 * JP * VS + JP * QS + 4 * JP * GS + 8;
constexpr size_t ROUND_LDS_BYTES = 88 * 1024;                  // > half of the CU's 160 KB: one workgroup per CU
static_assert(ROUND_LDS_DOUBLES * 8 + 1024 <= ROUND_LDS_BYTES, "LDS carve-up of the fused round kernel");
constexpr int RGMAX = 8;                                        // most parts per pair
constexpr int RPRE = 6;                                         // update chunks held in registers across the eigensolve

__device__ __forceinline__ int ring_next(int p)  // destination position of position p after one step
{
    if (p == 0) return 0;
    if (p == 1) return 2;
    if (p & 1) return p - 2;          // bottom row moves left
    return p == JP - 2 ? JP - 1 : p + 2; // top row moves right, its last place drops to the bottom row
}

// Branch-free form of jacobi_rot (same formulas): 1/sqrt by v_rsq_f64 + two coupled Goldschmidt steps that deliver
// sqrt and 1/(2 sqrt) together -- two dependent FMAs per step.  A thread derives its row AND its column rotation; without
// branches the two chains interleave (one wave per SIMD: every dependent f64 operation is ~16 exposed cycles).
// thr2: relative threshold -- a coupling with b^2 <= thr2 a d is left alone.  Without it two rows of EQUAL norm (a multiple
// singular value) are rotated by 45 degrees on a coupling of pure rounding noise, which drags the not-yet-annihilated couplings
// of one into the other at FIRST order: convergence in the presence of clusters turns linear, and the prediction of the last
// sweep (jacobi_orthogonalise) left 3e-10 ... 8e-10 of non-orthogonality behind on blocks with eight-fold values
// (scripts/svd_fuzz.py).  The callers pass (tol / 2)^2: such a pair counts as converged anyway.
__device__ __forceinline__ void jacobi_rot_bf(double a, double d, double b, double& c, double& s, double thr2 = 0.0)
{
    const double delta = d - a;
    const double b2 = b + b;
    double h2 = fma(b2, b2, delta * delta);
    const bool ok = (fabs(b) > 1e-300) & (h2 > 1e-300) & (b * b > thr2 * a * d);
    h2 = ok ? h2 : 1.0;
    double r = __builtin_amdgcn_rsq(h2);
    double g = h2 * r, h = 0.5 * r;
    double e = fma(-g, h, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-g, h, 0.5);
    h = fma(h, e, h);                             // 1 / (2 hyp)
    const double c2 = fma(fabs(delta), h, 0.5);   // cos^2 theta in [1/2, 1]
    double r2 = __builtin_amdgcn_rsq(c2);
    double gc = c2 * r2, hc = 0.5 * r2;
    double ec = fma(-gc, hc, 0.5);
    gc = fma(gc, ec, gc);
    hc = fma(hc, ec, hc);
    ec = fma(-gc, hc, 0.5);
    gc = fma(gc, ec, gc);                         // cos theta
    hc = fma(hc, ec, hc);                         // 1 / (2 cos theta)
    const double sabs = (fabs(b) * h) * (4.0 * hc); // |b| / (hyp cos theta)
    const bool pos = (delta >= 0.0) == (b >= 0.0);
    c = ok ? gc : 1.0;
    s = ok ? (pos ? sabs : -sabs) : 0.0;
}

__global__ void __launch_bounds__(NT, 1)
jacobi_round_kernel(const RPair* __restrict__ pairs, int round, int G, int max_inner,
                    unsigned long long* __restrict__ offmax_bits, RScratch sc)
{
    extern __shared__ __attribute__((aligned(16))) double rsm[];
    double* Gs = rsm;                    // Gram, buffer a
    double* G2 = Gs + JP * GS;           // Gram, buffer b
    double* Va = G2 + JP * GS;           // accumulated rotations, buffer a
    double* Vb = Va + JP * VS;
    double* Qs = Vb + JP * VS;           // Qm^T operand of the update: [k][m], output-row order
    double* Xc = Qs + JP * QS;           // wave slices of the partial Gram (4 x JP x GS)
    double* red = Xc + 4 * JP * GS;
    int* ibase = reinterpret_cast<int*>(red + 8);
    int* perm = ibase;                   // output row -> eigenvector column (descending eigenvalue)
    int* zrow = ibase + JP;              // 1: this row of the pair is numerically null (deflated)
    int* zout = ibase + 2 * JP;          // zrow in output-row order
    int* flags = ibase + 3 * JP;         // [0] any null  [1] wait ok

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int pi = blockIdx.x / G, part = blockIdx.x % G;
    // (diagnostic stamps go to a buffer nothing else reads; no output depends on them)
#define ROUND_STAMP(k)                                                                              \
    do {                                                                                            \
        if (sc.stamps && tid == 0) sc.stamps[(size_t)blockIdx.x * 8 + (k)] = wall_clock64();        \
    } while (0)
    ROUND_STAMP(0);
    const RPair mt = pairs[pi];
    if (round >= mt.nb - 1) return; // this matrix has fewer rounds per sweep
    int P, Q;
    circle_pair(mt.nb, round, mt.slot, P, Q);
    if (P > Q) {
        const int t = P;
        P = Q;
        Q = t;
    }
    // column shares in whole 64-column chunks
    const int cw = mt.lenp / 64, cj = mt.J ? mt.nvp / 64 : 0;
    const int w_begin = (int)((int64_t)part * cw / G), w_end = (int)((int64_t)(part + 1) * cw / G);
    const int j_begin = (int)((int64_t)part * cj / G), j_end = (int)((int64_t)(part + 1) * cj / G);
    const int nW = w_end - w_begin, nJ = j_end - j_begin;
    const int ct = nW + nJ;
    const int un = lane & 15, ukq = lane >> 4;
    auto chunk_ptr = [&](int c, int k) -> gp { // update layout: element (row k of the pair, column un of wave's 16) of own chunk c
        const bool in_w = c < nW;
        gp base = (gp)(in_w ? mt.W : mt.J);
        const int ld = in_w ? mt.lenp : mt.nvp;
        const int cc = in_w ? w_begin + c : j_begin + (c - nW);
        return base + (int64_t)xrow(k, P, Q) * ld + cc * 64 + wave * 16 + un;
    };

    // ---- 1. partial Gram over the own W share (see jacobi_gram_kernel for the operand mapping)
    double xpre[RPRE][JP / 4];
    {
        gp W = (gp)mt.W;
        const int ld = mt.lenp;
        d4 acc00 = d4{0.0, 0.0, 0.0, 0.0}, acc01 = acc00, acc11 = acc00;
        const int r = lane & 15, g2 = 2 * (lane >> 4);
        gcp2 s0 = (gcp2)(W + (int64_t)xrow(r, P, Q) * ld + g2);
        gcp2 s1 = (gcp2)(W + (int64_t)xrow(r + 16, P, Q) * ld + g2);
        const int c_begin = 8 * w_begin, c_end = 8 * w_end; // in 8-column chunks
        constexpr int UN = 8;
        d2 x0[UN], x1[UN], n0[UN], n1[UN];
        auto load = [&](d2 (&a0)[UN], d2 (&a1)[UN], int c0) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int c = min(c0 + 4 * u, c_end - 1);
                a0[u] = s0[c * 4];
                a1[u] = s1[c * 4];
            }
        };
        if (c_begin + wave < c_end) load(x0, x1, c_begin + wave);
        // the operands of step 4 (own share in update layout) are requested now and wait behind the eigensolve
#pragma unroll
        for (int c = 0; c < RPRE; ++c) {
            if (c < ct) {
#pragma unroll
                for (int kk = 0; kk < JP / 4; ++kk) xpre[c][kk] = *chunk_ptr(c, 4 * kk + ukq);
            }
        }
        for (int c0 = c_begin + wave; c0 < c_end; c0 += 4 * UN) {
            const bool more = c0 + 4 * UN < c_end;
            if (more) load(n0, n1, c0 + 4 * UN);
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (c0 + 4 * u < c_end) {
                    acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].x, x0[u].x, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].x, x1[u].x, acc01, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u].x, x1[u].x, acc11, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].y, x0[u].y, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].y, x1[u].y, acc01, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u].y, x1[u].y, acc11, 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    x0[u] = n0[u];
                    x1[u] = n1[u];
                }
            }
        }
        double* wpart = Xc + wave * (JP * GS);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = (lane >> 4) + 4 * q, cc = lane & 15;
            wpart[rr * GS + cc] = acc00[q];
            wpart[rr * GS + 16 + cc] = acc01[q];
            wpart[(16 + cc) * GS + rr] = acc01[q]; // mirror of tile (0,1)
            wpart[(16 + rr) * GS + 16 + cc] = acc11[q];
        }
    }
    __syncthreads();
    ROUND_STAMP(1);
    {
        // every thread owns four consecutive entries of the 32 x 32 matrix (row e4 / 8, columns 4 (e4 % 8) ...)
        const int gi = tid >> 3, gj = 4 * (tid & 7);
        d2 lo = d2{0.0, 0.0}, hi = d2{0.0, 0.0};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const double* src = Xc + w * (JP * GS) + gi * GS + gj;
            lo.x += src[0];
            lo.y += src[1];
            hi.x += src[2];
            hi.y += src[3];
        }
        if (G > 1) {
            // ---- 2. exchange of the partials among the G parts of this pair
            const unsigned int pbytes = JP * JP * 8;
            __amdgpu_buffer_rsrc_t rs_all = __builtin_amdgcn_make_buffer_rsrc(sc.gpart + (size_t)pi * G * (JP * JP), 0, pbytes * G, 0x00020000);
            Q16 a, b;
            a.d = lo;
            b.d = hi;
            const unsigned int off = (unsigned int)part * pbytes + (unsigned int)tid * 32u;
            __builtin_amdgcn_raw_buffer_store_b128(a.u, rs_all, off, 0, 16);        // aux 16 = sc1: write-through
            __builtin_amdgcn_raw_buffer_store_b128(b.u, rs_all, off + 16u, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains ...
            __syncthreads();                                  // ... before ONE lane takes the ticket for the workgroup
            if (tid == 0) {
                unsigned int* ticket = sc.cnt + (size_t)round * sc.np + pi;
                __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int ok = 1;
                const unsigned long long t0 = wall_clock64(); // 100 MHz
                while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)G) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > 100000000ull) { // one second: a partner is not resident / died
                        ok = 0;
                        __hip_atomic_store(sc.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                flags[1] = ok;
            }
            __syncthreads(); // the polling wave joins the barrier after its poll matched; the others load after it
            if (!flags[1]) return;
            Q16 pa[RGMAX], pb[RGMAX];
#pragma unroll
            for (int g = 0; g < RGMAX; ++g) { // every load in flight before the first use
                if (g < G) {
                    pa[g].u = __builtin_amdgcn_raw_buffer_load_b128(rs_all, (unsigned int)g * pbytes + (unsigned int)tid * 32u, 0, 16);
                    pb[g].u = __builtin_amdgcn_raw_buffer_load_b128(rs_all, (unsigned int)g * pbytes + (unsigned int)tid * 32u + 16u, 0, 16);
                }
            }
            lo = d2{0.0, 0.0};
            hi = d2{0.0, 0.0};
#pragma unroll
            for (int g = 0; g < RGMAX; ++g) { // the same order in every part: bit-identical sums
                if (g < G) {
                    lo += pa[g].d;
                    hi += pb[g].d;
                }
            }
        }
        double* dst = Gs + gi * GS + gj;
        dst[0] = lo.x;
        dst[1] = lo.y;
        dst[2] = hi.x;
        dst[3] = hi.y;
    }
    __syncthreads();
    ROUND_STAMP(2);
    // ---- 3a. deflation: rows whose squared norm fell below the numerical-rank threshold are zeroed
    if (tid == 0) flags[0] = 0;
    __syncthreads();
    if (tid < JP) {
        const double g = Gs[tid * GS + tid];
        const int z = (g > 0.0 && g <= mt.thr2) ? 1 : 0;
        zrow[tid] = z;
        perm[tid] = tid;
        if (z) flags[0] = 1;
    }
    __syncthreads();
    const bool any_null = flags[0] != 0;
    if (any_null) {
        for (int e = tid; e < JP * JP; e += NT) {
            const int i = e / JP, j = e % JP;
            if (zrow[i] || zrow[j]) Gs[i * GS + j] = 0.0;
        }
        __syncthreads();
    }
    // ---- 3b. convergence measure; nothing to do if this pair is already orthogonal
    const double off = gram_offmax(Gs, red, tid);
    if (part == 0 && tid == 0) atomicMax(offmax_bits + mt.mat, (unsigned long long)__double_as_longlong(off));
    if (off <= mt.tol && !any_null) return; // (the same decision in every part: identical Gram matrices)

    ROUND_STAMP(3);
    // ---- 3c. two-sided Jacobi eigensolve in position space
    for (int e = tid; e < JP * VS; e += NT) Va[e] = ((e / VS) == (e % VS)) ? 1.0 : 0.0;
    __syncthreads();
    double* Ga = Gs;
    double* Gb = G2;
    double* Vc = Va;
    double* Vn = Vb;
    {
        const int pr = tid >> 4, pc = tid & 15;
        const int r0 = 2 * pr, c0 = 2 * pc;
        const int dr0 = ring_next(r0), dr1 = ring_next(r0 + 1), dc0 = ring_next(c0), dc1 = ring_next(c0 + 1);
        for (int sweep = 0; sweep < (off <= mt.tol ? 0 : max_inner); ++sweep) {
            for (int r = 0; r < JP - 1; ++r) {
                const d2 g0 = *reinterpret_cast<const d2*>(Ga + r0 * GS + c0);
                const d2 g1 = *reinterpret_cast<const d2*>(Ga + (r0 + 1) * GS + c0);
                const d2 ac = *reinterpret_cast<const d2*>(Ga + c0 * GS + c0);
                const double dc = Ga[(c0 + 1) * GS + c0 + 1];
                const d2 v0 = *reinterpret_cast<const d2*>(Vc + r0 * VS + c0);
                const d2 v1 = *reinterpret_cast<const d2*>(Vc + (r0 + 1) * VS + c0);
                // ONE rotation chain per thread (its column pair); the row pair's rotation comes from lane pc == pr of the
                // same 16-lane row (the step is bound by the f64 instruction count: 1196 -> 935 cycles, eig_step_probe.hip)
                double c1, s1, c2, s2;
                jacobi_rot_bf(ac.x, dc, ac.y, c2, s2);
                const int rsrc = (lane & 48) | pr;
                c1 = __shfl(c2, rsrc);
                s1 = __shfl(s2, rsrc);
                // G <- R1^T G R2 on the own 2x2 block
                const double hik = c1 * g0.x - s1 * g1.x, hil = c1 * g0.y - s1 * g1.y;
                const double hjk = s1 * g0.x + c1 * g1.x, hjl = s1 * g0.y + c1 * g1.y;
                double nik = c2 * hik - s2 * hil, nil = s2 * hik + c2 * hil;
                double njk = c2 * hjk - s2 * hjl, njl = s2 * hjk + c2 * hjl;
                if (pr == pc) { // the rotated diagonal block is diagonal by construction
                    nil = 0.0;
                    njk = 0.0;
                }
                Gb[dr0 * GS + dc0] = nik;
                Gb[dr0 * GS + dc1] = nil;
                Gb[dr1 * GS + dc0] = njk;
                Gb[dr1 * GS + dc1] = njl;
                // V <- V R2 : rows (r0, r0 + 1) stay, column positions move with the ring
                Vn[r0 * VS + dc0] = c2 * v0.x - s2 * v0.y;
                Vn[r0 * VS + dc1] = s2 * v0.x + c2 * v0.y;
                Vn[(r0 + 1) * VS + dc0] = c2 * v1.x - s2 * v1.y;
                Vn[(r0 + 1) * VS + dc1] = s2 * v1.x + c2 * v1.y;
                __syncthreads();
                double* t = Ga;
                Ga = Gb;
                Gb = t;
                t = Vc;
                Vc = Vn;
                Vn = t;
            }
            if (sweep + 1 < max_inner) {
                const double off_in = gram_offmax(Ga, red, tid);
                if (off_in <= 0.25 * mt.tol) break;
            }
        }
    }
    __syncthreads();
    ROUND_STAMP(4);
    // de Rijk-style ordering inside the pair: larger norms to the lower rows; padding rows stay last (see jacobi_gram_kernel)
    if (off > mt.tol && tid < JP) {
        auto key = [&](int i) { return xrow(i, P, Q) < mt.nv ? Ga[i * GS + i] : -1.0e300; };
        const double g = key(tid);
        int rk = 0;
        for (int j = 0; j < JP; ++j) {
            const double gj = key(j);
            rk += (gj > g || (gj == g && j < tid)) ? 1 : 0;
        }
        perm[rk] = tid;
    }
    __syncthreads();
    // ---- 4. X <- Qm^T X on the own chunks.  A operand of the MFMA: A[m][k] = Qm[k][perm[m]], kept as [k][m]
    for (int e = tid; e < JP * JP; e += NT) {
        const int k = e / JP, m = e % JP;
        Qs[k * QS + m] = Vc[k * VS + perm[m]];
    }
    if (tid < JP) zout[tid] = zrow[perm[tid]];
    __syncthreads();
    ROUND_STAMP(5);
    if (ct <= 0) return;
    const double* ap0 = Qs + (lane >> 4) * QS + (lane & 15);
    const double* ap1 = ap0 + 16;
    double a0[JP / 4], a1[JP / 4];
#pragma unroll
    for (int kk = 0; kk < JP / 4; ++kk) {
        a0[kk] = ap0[kk * 4 * QS];
        a1[kk] = ap1[kk * 4 * QS];
    }
    int zr[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) zr[i][q] = zout[i * 16 + (lane >> 4) + 4 * q];
    auto apply = [&](int c, const double (&xb)[JP / 4]) {
        d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int kk = 0; kk < JP / 4; ++kk) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], xb[kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], xb[kk], acc[1], 0, 0, 0);
        }
        const bool zero_null = c < nW; // deflated rows are zeroed in W only
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = i * 16 + (lane >> 4) + 4 * q;
                *chunk_ptr(c, row) = (zero_null && zr[i][q]) ? 0.0 : acc[i][q];
            }
    };
#pragma unroll
    for (int c = 0; c < RPRE; ++c)
        if (c < ct) apply(c, xpre[c]);
    if (ct > RPRE) { // long shares (few parts per pair): the rest streams with one chunk of prefetch
        double xb[JP / 4], xn[JP / 4];
#pragma unroll
        for (int kk = 0; kk < JP / 4; ++kk) xb[kk] = *chunk_ptr(RPRE, 4 * kk + ukq);
        for (int c = RPRE; c < ct; ++c) {
            if (c + 1 < ct) {
#pragma unroll
                for (int kk = 0; kk < JP / 4; ++kk) xn[kk] = *chunk_ptr(c + 1, 4 * kk + ukq);
            }
            apply(c, xb);
#pragma unroll
            for (int kk = 0; kk < JP / 4; ++kk) xb[kk] = xn[kk];
        }
    }
    __syncthreads();
    ROUND_STAMP(6);
#undef ROUND_STAMP
}


// ================================================================================================
// Persistent sweep: ONE launch per sweep (all rounds of the round-robin schedule of every matrix).
//
// Workgroup (pair slot s of matrix m, column part g) stays on its CU for the whole sweep and walks the rounds
// r = 0 .. nb_m - 2 itself.  What used to be a kernel boundary between two rounds is a DATAFLOW hand-off: the rows of
// block P (column share g) updated in round r by the workgroup that held P then are picked up in round r + 1 by the
// workgroup whose slot the schedule moves P to -- same part g, so every (block, part) piece has exactly one owner per
// round and ownership travels with a monotonic counter  ready[block][part] = rounds completed  (Guideline 16: the piece
// is stored write-through (sc1), every storing wave drains, barrier, ONE lane publishes the counter; the next owner polls
// it with ONE lane, barrier, then reads the piece with sc1 loads).  Consequences:
//   * no launch ramp / drain / descriptor fetch per round (8-10 us of a 44 us fused round);
//   * matrices are DECOUPLED: a small block never waits for a large one and vice versa -- the lockstep of the
//     launch-per-round paths made every round as slow as its slowest pair and kept the small blocks' pairs (most of the
//     pairs of a theta list) in the large blocks' rounds;
//   * parts per pair are chosen per matrix (long chains get the CUs): G_m in RPair.pad[0].
// The partial-Gram exchange among the G parts of a pair is the one of jacobi_round_kernel, with the partials double
// buffered by round parity (a part can be at most one exchange ahead of its partners) and a monotonic ticket
// (target G (r + 1)).  Every wait is bounded by the wall clock AND leaves as soon as any workgroup has raised the error
// word, so a lost partner turns into an error code within a second, never into a hang.  All workgroups must be resident:
// the host launches at most one per CU and the kernel's LDS request admits no second one.
// ---------------------------------------------------------------------------------------------
// Structure-preserving pivot solve for the interleaved real embedding of COMPLEX rows: the
// 32 x 32 Gram matrix of a pair of 16-row blocks -- 8 + 8 complex rows, real row 2a (+1) = real (imaginary) embedding row of
// complex row a -- is M(G_c) of a 16 x 16 Hermitian G_c.  One wave diagonalises G_c by cyclic complex Jacobi rotations in
// LDS and leaves M(Q_c) and M(Lambda): the real update that follows then IS the complex one, rows stay structured and every
// singular value appears once per complex row.  A rotation for the Hermitian 2 x 2 [[a, g], [conj g, d]] is the real Jacobi
// rotation of [[a, |g|], [|g|, d]] behind the phase g / |g|:  U = [[c, s], [-s conj(phi), c conj(phi)]].
constexpr int CJ = JP / 2;  // complex rows per pair problem
constexpr int CS = CJ + 1;  // row stride of the complex work arrays
__device__ __forceinline__ void hermitian_pivot_solve(const double* Gs, double* work, double* rot, double* Vout, double* Gdiag, int sweeps,
                                                      int lane, double thr2 = 0.0, bool cross = false)
{   // work: 4 * CJ * CS doubles;  rot: 2 * CJ doubles (CJ/2 rotations x c, s, phi_re, phi_im)
    // cross: only the CJ/2 rotation sets that pair a complex row of block P (0 .. CJ/2 - 1) with one of block Q (SScratch::cross_every)
    auto pair_of = [cross](int r, int k, int& p, int& q) {
        if (cross) {
            p = k;
            q = CJ / 2 + ((k + r) & (CJ / 2 - 1));
        } else
            circle_pair(CJ, r, k, p, q);
    };
    const int n_steps = cross ? CJ / 2 : CJ - 1;
    double* Hr = work;                 // [CJ][CS] each
    double* Hi = Hr + CJ * CS;
    double* Ur = Hi + CJ * CS;
    double* Ui = Ur + CJ * CS;
    for (int e = lane; e < CJ * CJ; e += 64) {
        const int a = e / CJ, b = e % CJ;
        // M(G_c)[2a][2b] = Re, M(G_c)[2a+1][2b] = Im; the two copies of every entry differ by rounding: average them
        Hr[a * CS + b] = 0.5 * (Gs[(2 * a) * GS + 2 * b] + Gs[(2 * a + 1) * GS + 2 * b + 1]);
        Hi[a * CS + b] = a == b ? 0.0 : 0.5 * (Gs[(2 * a + 1) * GS + 2 * b] - Gs[(2 * a) * GS + 2 * b + 1]);
        Ur[a * CS + b] = a == b ? 1.0 : 0.0;
        Ui[a * CS + b] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int sw = 0; sw < sweeps; ++sw)
        for (int r = 0; r < n_steps; ++r) {
            if (lane < CJ / 2) {
                int p, q;
                pair_of(r, lane, p, q);
                const double gr = Hr[p * CS + q], gi = Hi[p * CS + q];
                const double ab = sqrt(gr * gr + gi * gi);
                double c = 1.0, sn = 0.0, pr = 1.0, pi = 0.0;
                if (ab > 1e-300) {
                    jacobi_rot_bf(Hr[p * CS + p], Hr[q * CS + q], ab, c, sn, thr2);
                    pr = gr / ab;
                    pi = gi / ab;
                }
                rot[lane * 4 + 0] = c;
                rot[lane * 4 + 1] = sn;
                rot[lane * 4 + 2] = pr;
                rot[lane * 4 + 3] = pi;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // rows:  row_p <- c row_p - s phi row_q,   row_q <- s row_p + c phi row_q        (H <- U^H H)
            for (int e = lane; e < (CJ / 2) * CJ; e += 64) {
                const int k = e / CJ, col = e % CJ;
                int p, q;
                pair_of(r, k, p, q);
                const double c = rot[k * 4], sn = rot[k * 4 + 1], fr = rot[k * 4 + 2], fi = rot[k * 4 + 3];
                const double pr_ = Hr[p * CS + col], pi_ = Hi[p * CS + col], qr_ = Hr[q * CS + col], qi_ = Hi[q * CS + col];
                const double tr = fr * qr_ - fi * qi_, ti = fr * qi_ + fi * qr_; // phi * row_q
                Hr[p * CS + col] = c * pr_ - sn * tr;
                Hi[p * CS + col] = c * pi_ - sn * ti;
                Hr[q * CS + col] = sn * pr_ + c * tr;
                Hi[q * CS + col] = sn * pi_ + c * ti;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // columns of H and of U:  col_p <- c col_p - s conj(phi) col_q,   col_q <- s col_p + c conj(phi) col_q   (. <- . U)
            for (int e = lane; e < (CJ / 2) * CJ * 2; e += 64) {
                const int which = e / ((CJ / 2) * CJ), f = e % ((CJ / 2) * CJ);
                const int k = f / CJ, row = f % CJ;
                int p, q;
                pair_of(r, k, p, q);
                double* Xr = which ? Ur : Hr;
                double* Xi = which ? Ui : Hi;
                const double c = rot[k * 4], sn = rot[k * 4 + 1], fr = rot[k * 4 + 2], fi = -rot[k * 4 + 3];
                const double pr_ = Xr[row * CS + p], pi_ = Xi[row * CS + p], qr_ = Xr[row * CS + q], qi_ = Xi[row * CS + q];
                const double tr = fr * qr_ - fi * qi_, ti = fr * qi_ + fi * qr_; // conj(phi) * col_q
                Xr[row * CS + p] = c * pr_ - sn * tr;
                Xi[row * CS + p] = c * pi_ - sn * ti;
                Xr[row * CS + q] = sn * pr_ + c * tr;
                Xi[row * CS + q] = sn * pi_ + c * ti;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    // expand: Vout[k][j] (row stride VS) = M(U)[k][j],  Gdiag[i][i] (row stride GS) = lambda_{i / 2}
    for (int e = lane; e < CJ * CJ; e += 64) {
        const int a = e / CJ, b = e % CJ;
        const double x = Ur[a * CS + b], y = Ui[a * CS + b];
        Vout[(2 * a) * VS + 2 * b] = x;
        Vout[(2 * a) * VS + 2 * b + 1] = -y;
        Vout[(2 * a + 1) * VS + 2 * b] = y;
        Vout[(2 * a + 1) * VS + 2 * b + 1] = x;
    }
    if (lane < CJ) {
        const double lam = Hr[lane * CS + lane];
        Gdiag[(2 * lane) * GS + 2 * lane] = lam;
        Gdiag[(2 * lane + 1) * GS + 2 * lane + 1] = lam;
    }
}

struct SScratch {
    double* gpart;            // [pair][2][G_max][JP*JP]
    unsigned int* ticket;     // [pair]                 arrivals of the Gram exchange (zeroed per sweep)
    unsigned int* ready;      // [matrix flag base + block * G + part]   rounds completed (zeroed per sweep)
    unsigned int* err;        // [1]
    const int2* wgmap;        // [entry] -> (pair, part)
    const int2* wgent;        // [workgroup] -> [begin, end) of its entries in wgmap (null: workgroup b owns entry b).  More than
                              // one entry per workgroup: lists with more pairs than the chip has CUs -- a workgroup then runs its
                              // entries one after the other in every round (short matrices share, long chains stay alone)
    unsigned long long* stamps;
    int stamp_round;
    // all sweeps in ONE launch (n_sweeps > 1): the convergence test runs on the device.  offmax_bits is then [sweep][matrix];
    // arrive[matrix] counts the (pair, part) entries of a matrix that have finished a sweep (monotonic), nsw[matrix] is the
    // number of sweeps the matrix took (0: not converged within n_sweeps)
    int n_sweeps, n_mats;
    unsigned int* arrive;
    int* nsw;
    // Deferred J update: a workgroup applies a round's rotation to its W share at once (the next owners wait for that), and to
    // its J share only while it waits for the partial Grams of the NEXT round -- nobody reads J before the end, so that half of
    // the update fills what was idle time.  readyJ[block][part]: rounds whose J update (or skip) is complete, as `ready` for W.
    unsigned int* readyJ;
    int defer_j;
    // Cross-only pivot solves (real engine, one inner sweep): the 31 rotation sets of a pivot solve are 16 sets of pairs (row of
    // block P, row of block Q) and 15 sets of pairs inside a block.  A sweep needs every pair of rows ONCE; the pairs inside a
    // block are met again in every round the block takes part in.  With cross_every = k > 0 only the rounds with round % k == 0
    // run all 31 sets, the others the 16 cross sets (positions interleaved P0 Q0 P1 Q1 ..., the Q side moves one place per step).
    int cross_every, cross_min_nb;
};
constexpr int SW_MAX_ENT = 64; // entries per workgroup the one-launch form keeps a done flag for

constexpr int SW_UN = 8;                                        // 8-column Gram chunks in flight per wave and buffer
constexpr int SW_PRE = 6;                                       // update chunks held in registers across the eigensolve
// [Gs | region shared by the wave slices of the partial Gram (Xc) and the eigensolver's G2 / Va / Vb | Qm of this round and of the
//  previous one (the deferred J update still reads it while the next Gram runs) | reduction scratch]
constexpr int SWEEP_XREG = (4 * JP * GS > JP * GS + 2 * JP * VS) ? 4 * JP * GS : JP * GS + 2 * JP * VS;
constexpr int SWEEP_LDS_DOUBLES = JP * GS + SWEEP_XREG + 2 * JP * QS + 8;
// ONE workgroup per CU (the LDS request is padded past half a CU's 160 KB).  Two per CU were measured and dropped: with the
// 256-register budget that needs (SW_UN = 2, SW_PRE = 2: no spills) and twice the parts per pair, a round of the largest
// block of the chi=4096 list took 44.5 us instead of 46 (eigensolve 19.5 instead of 16.7 us with a second wave on every
// SIMD, exchange among 8 parts 11 instead of 6.5 us) and the batched SVD 52.3 instead of 49.9 ms.
constexpr size_t SWEEP_LDS_BYTES = 88 * 1024;
static_assert((size_t)SWEEP_LDS_DOUBLES * 8 + 1024 <= SWEEP_LDS_BYTES, "LDS request of the sweep kernel");
constexpr int SWEEP_WG_PER_CU = 1;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
union Q8 {
    u32x2 u;
    double d;
};

template <int SLEEP = 1>
__device__ __forceinline__ bool spin_until(const unsigned int* word, unsigned int target, unsigned int* err)
{
    const unsigned long long t0 = wall_clock64(); // 100 MHz
    unsigned int it = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if ((++it & 63u) == 0u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (wall_clock64() - t0 > 100000000ull) { // one second
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

// J rows of the pair (P, Q), column chunks [j_begin, j_begin + nJ) of 64:  X <- Qm^T X  (Qm in LDS as [k][m], stride QS)
__device__ __forceinline__ void sweep_update_j(const __amdgpu_buffer_rsrc_t rsJ, const double* Qm, int P, int Q, int nvp, int j_begin,
                                               int nJ, int wave, int lane)
{
    const int un = lane & 15, ukq = lane >> 4;
    const double* ap0 = Qm + (lane >> 4) * QS + (lane & 15);
    const double* ap1 = ap0 + 16;
    double a0[JP / 4], a1[JP / 4];
#pragma unroll
    for (int kk = 0; kk < JP / 4; ++kk) {
        a0[kk] = ap0[kk * 4 * QS];
        a1[kk] = ap1[kk * 4 * QS];
    }
    auto off_of = [&](int c, int k) { return (unsigned int)(((int64_t)xrow(k, P, Q) * nvp + (j_begin + c) * 64 + wave * 16 + un) * 8); };
    auto load = [&](double (&x)[JP / 4], int c) {
#pragma unroll
        for (int kk = 0; kk < JP / 4; ++kk) {
            Q8 q;
            q.u = __builtin_amdgcn_raw_buffer_load_b64(rsJ, off_of(c, 4 * kk + ukq), 0, 16);
            x[kk] = q.d;
        }
    };
    double xb[JP / 4], xn[JP / 4];
    if (nJ > 0) load(xb, 0);
    for (int c = 0; c < nJ; ++c) {
        if (c + 1 < nJ) load(xn, c + 1);
        d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int kk = 0; kk < JP / 4; ++kk) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], xb[kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], xb[kk], acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Q8 v;
                v.d = acc[i][q];
                __builtin_amdgcn_raw_buffer_store_b64(v.u, rsJ, off_of(c, i * 16 + (lane >> 4) + 4 * q), 0, 16);
            }
#pragma unroll
        for (int kk = 0; kk < JP / 4; ++kk) xb[kk] = xn[kk];
    }
}

template <bool CPLX>
__global__ void __launch_bounds__(NT, 1)
jacobi_sweep_kernel(const RPair* __restrict__ pairs, int max_inner, unsigned long long* __restrict__ offmax_bits, SScratch sc)
{
    extern __shared__ __attribute__((aligned(16))) double rsm[];
    // LDS carve-up for two to three workgroups per CU: the wave slices of the partial Gram (Xc) are dead once the
    // exchange has produced Gs, so they share their space with the eigensolver's second Gram buffer, V and Qs
    double* Gs = rsm;
    double* G2 = Gs + JP * GS;
    double* Va = G2 + JP * GS;
    double* Vb = Va + JP * VS;
    double* Xc = G2;                                        // (SWEEP_XREG doubles: the larger of the two uses)
    double* Qs0 = G2 + SWEEP_XREG;                          // Qm of even rounds ...
    double* Qs1 = Qs0 + JP * QS;                            // ... and of odd ones (counted per matrix: `ground`)
    double* red = Qs1 + JP * QS;
    int* ibase = reinterpret_cast<int*>(red + 8);
    int* perm = ibase;
    int* zrow = ibase + JP;
    int* zout = ibase + 2 * JP;
    int* flags = ibase + 3 * JP; // [0] any null  [1] wait ok  [2] every entry of this workgroup has converged
    int* edone = flags + 4;      // [SW_MAX_ENT] one-launch form: entry has converged

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int2 went = sc.wgent ? sc.wgent[blockIdx.x] : make_int2((int)blockIdx.x, (int)blockIdx.x + 1);
    int max_rounds = 0;
    for (int e = went.x; e < went.y; ++e) max_rounds = max(max_rounds, pairs[sc.wgmap[e].x].nb - 1);
    const int n_sw = sc.n_sweeps;
    // deferred J update (workgroups with ONE entry: pair slot and part are then the same in every round)
    const bool one_entry = went.y - went.x == 1;
    int pendP = -1, pendQ = -1, pend_ground = 0; // J update of an earlier round still to be applied (pendP < 0: none)
    if (n_sw > 1) {
        if (tid < SW_MAX_ENT) edone[tid] = 0;
        __syncthreads();
    }
    for (int sw = 0; sw < n_sw; ++sw) {
    for (int round = 0; round < max_rounds; ++round)
    for (int ent = went.x; ent < went.y; ++ent) {
    const int2 wm = sc.wgmap[ent];
    const int pi = wm.x, part = wm.y;
    const RPair mt = pairs[pi];
    if (round >= mt.nb - 1) continue; // (workgroup-uniform)
    if (n_sw > 1 && edone[ent - went.x]) continue;
    const int ground = sw * (mt.nb - 1) + round; // rounds this matrix has been through (counters and buffers never reset)
    const int G = mt.pad[0];
    unsigned int* ready = sc.ready + mt.pad[1];
    unsigned int* readyJ = sc.readyJ + mt.pad[1];
    unsigned int* ticket = sc.ticket + pi;
    double* Qs = (ground & 1) ? Qs1 : Qs0;
    const unsigned int pbytes = JP * JP * 8;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(mt.W, 0, (int)((size_t)mt.nvp * mt.lenp * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsJ =
        __builtin_amdgcn_make_buffer_rsrc(mt.J ? mt.J : mt.W, 0, mt.J ? (int)((size_t)mt.nvp * mt.nvp * 8) : 0, 0x00020000);
    // column shares in whole 64-column chunks (fixed for the sweep)
    const int cw = mt.lenp / 64, cj = mt.J ? mt.nvp / 64 : 0;
    const int w_begin = (int)((int64_t)part * cw / G), w_end = (int)((int64_t)(part + 1) * cw / G);
    const int j_begin = (int)((int64_t)part * cj / G), j_end = (int)((int64_t)(part + 1) * cj / G);
    const int nW = w_end - w_begin, nJ = j_end - j_begin;
    // this round's J half is deferred into the exchange wait of the next round (needs an exchange: G > 1)
    const bool defer = sc.defer_j && one_entry && G > 1 && nJ > 0 && round < mt.nb - 2; // (the last round of a sweep updates J at once)
    const int ct = defer ? nW : nW + nJ;
    const int pre_limit = (sc.defer_j && G > 1) ? min(ct, nW) : ct; // chunks that may be prefetched before the eigensolve
    const int un = lane & 15, ukq = lane >> 4;
#define SWEEP_STAMP(k)                                                                                            \
    do {                                                                                                          \
        if (sc.stamps && tid == 0 && round == sc.stamp_round) sc.stamps[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); \
    } while (0)

    {
        int P, Q;
        circle_pair(mt.nb, round, mt.slot, P, Q);
        if (P > Q) {
            const int t = P;
            P = Q;
            Q = t;
        }
        SWEEP_STAMP(0);
        // ---- 0. take over the two row blocks: their previous owners have finished round - 1
        __syncthreads(); // (also: nobody still reads the LDS of the previous round)
        if (ground > 0) {
            if (tid == 0) {
                bool ok = spin_until(ready + (size_t)P * G + part, (unsigned int)ground, sc.err);
                ok = ok && spin_until(ready + (size_t)Q * G + part, (unsigned int)ground, sc.err);
                flags[1] = ok ? 1 : 0;
            }
            __syncthreads();
            if (!flags[1]) return;
        }
        SWEEP_STAMP(1);
        // byte offset of element (row k of the pair, own chunk c, column un of this wave's 16) and its descriptor
        auto chunk_off = [&](int c, int k, bool& in_w) -> unsigned int {
            in_w = c < nW;
            const int ld = in_w ? mt.lenp : mt.nvp;
            const int cc = in_w ? w_begin + c : j_begin + (c - nW);
            return (unsigned int)(((int64_t)xrow(k, P, Q) * ld + cc * 64 + wave * 16 + un) * 8);
        };
        auto chunk_load = [&](int c, int k) -> double {
            bool in_w;
            const unsigned int off = chunk_off(c, k, in_w);
            Q8 q;
            q.u = in_w ? __builtin_amdgcn_raw_buffer_load_b64(rsW, off, 0, 16) : __builtin_amdgcn_raw_buffer_load_b64(rsJ, off, 0, 16);
            return q.d;
        };
        // ---- 1. partial Gram over the own W share
        double xpre[SW_PRE][JP / 4];
        {
            const int ld = mt.lenp;
            d4 acc00 = d4{0.0, 0.0, 0.0, 0.0}, acc01 = acc00, acc11 = acc00;
            const int r = lane & 15, g2 = 2 * (lane >> 4);
            const unsigned int o0 = (unsigned int)(((int64_t)xrow(r, P, Q) * ld + g2) * 8);
            const unsigned int o1 = (unsigned int)(((int64_t)xrow(r + 16, P, Q) * ld + g2) * 8);
            const int c_begin = 8 * w_begin, c_end = 8 * w_end; // in 8-column chunks
            constexpr int UN = SW_UN;
            d2 x0[UN], x1[UN], n0[UN], n1[UN];
            auto load = [&](d2 (&a0)[UN], d2 (&a1)[UN], int c0) {
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int c = min(c0 + 4 * u, c_end - 1);
                    Q16 qa, qb;
                    qa.u = __builtin_amdgcn_raw_buffer_load_b128(rsW, o0 + (unsigned int)c * 64u, 0, 16);
                    qb.u = __builtin_amdgcn_raw_buffer_load_b128(rsW, o1 + (unsigned int)c * 64u, 0, 16);
                    a0[u] = qa.d;
                    a1[u] = qb.d;
                }
            };
            if (c_begin + wave < c_end) load(x0, x1, c_begin + wave);
#pragma unroll
            for (int c = 0; c < SW_PRE; ++c) {
                if (c < pre_limit) { // (J chunks of a matrix whose J updates may be deferred are loaded after the wait for them)
#pragma unroll
                    for (int kk = 0; kk < JP / 4; ++kk) xpre[c][kk] = chunk_load(c, 4 * kk + ukq);
                }
            }
            for (int c0 = c_begin + wave; c0 < c_end; c0 += 4 * UN) {
                const bool more = c0 + 4 * UN < c_end;
                if (more) load(n0, n1, c0 + 4 * UN);
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    if (c0 + 4 * u < c_end) {
                        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].x, x0[u].x, acc00, 0, 0, 0);
                        acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].x, x1[u].x, acc01, 0, 0, 0);
                        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u].x, x1[u].x, acc11, 0, 0, 0);
                        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].y, x0[u].y, acc00, 0, 0, 0);
                        acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u].y, x1[u].y, acc01, 0, 0, 0);
                        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u].y, x1[u].y, acc11, 0, 0, 0);
                    }
                }
                if (more) {
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        x0[u] = n0[u];
                        x1[u] = n1[u];
                    }
                }
            }
            double* wpart = Xc + wave * (JP * GS);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = (lane >> 4) + 4 * q, cc = lane & 15;
                wpart[rr * GS + cc] = acc00[q];
                wpart[rr * GS + 16 + cc] = acc01[q];
                wpart[(16 + cc) * GS + rr] = acc01[q];
                wpart[(16 + rr) * GS + 16 + cc] = acc11[q];
            }
        }
        __syncthreads();
        SWEEP_STAMP(2);
        {
            const int gi = tid >> 3, gj = 4 * (tid & 7);
            d2 lo = d2{0.0, 0.0}, hi = d2{0.0, 0.0};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const double* src = Xc + w * (JP * GS) + gi * GS + gj;
                lo.x += src[0];
                lo.y += src[1];
                hi.x += src[2];
                hi.y += src[3];
            }
            if (G > 1) {
                // ---- 2. exchange of the partials among the G parts (double buffered by round parity)
                const __amdgpu_buffer_rsrc_t rs_all = __builtin_amdgcn_make_buffer_rsrc(
                    sc.gpart + ((size_t)pi * 2 + (size_t)(ground & 1)) * RGMAX * (JP * JP), 0, (int)(pbytes * G), 0x00020000);
                Q16 a, b;
                a.d = lo;
                b.d = hi;
                const unsigned int off = (unsigned int)part * pbytes + (unsigned int)tid * 32u;
                __builtin_amdgcn_raw_buffer_store_b128(a.u, rs_all, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(b.u, rs_all, off + 16u, 0, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pendP >= 0) {
                    // ---- the J half of the previous round's update, while the partners' partial Grams are on their way
                    if (tid == 0) {
                        bool ok = spin_until(readyJ + (size_t)pendP * G + part, (unsigned int)pend_ground, sc.err);
                        ok = ok && spin_until(readyJ + (size_t)pendQ * G + part, (unsigned int)pend_ground, sc.err);
                        flags[1] = ok ? 1 : 0;
                    }
                    __syncthreads();
                    if (!flags[1]) return;
                    sweep_update_j(rsJ, (pend_ground & 1) ? Qs1 : Qs0, pendP, pendQ, mt.nvp, j_begin, nJ, wave, lane);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) {
                        __hip_atomic_store(readyJ + (size_t)pendP * G + part, (unsigned int)(pend_ground + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(readyJ + (size_t)pendQ * G + part, (unsigned int)(pend_ground + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    pendP = -1;
                }
                if (tid == 0) flags[1] = spin_until(ticket, (unsigned int)G * (unsigned int)(ground + 1), sc.err) ? 1 : 0;
                __syncthreads();
                if (!flags[1]) return;
                lo = d2{0.0, 0.0};
                hi = d2{0.0, 0.0};
                for (int g0 = 0; g0 < G; g0 += 4) { // four parts' loads in flight at a time; the same order in every part
                    Q16 pa[4], pb[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g0 + g < G) {
                            pa[g].u = __builtin_amdgcn_raw_buffer_load_b128(rs_all, (unsigned int)(g0 + g) * pbytes + (unsigned int)tid * 32u, 0, 16);
                            pb[g].u = __builtin_amdgcn_raw_buffer_load_b128(rs_all, (unsigned int)(g0 + g) * pbytes + (unsigned int)tid * 32u + 16u, 0, 16);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g0 + g < G) {
                            lo += pa[g].d;
                            hi += pb[g].d;
                        }
                    }
                }
            }
            double* dst = Gs + gi * GS + gj;
            dst[0] = lo.x;
            dst[1] = lo.y;
            dst[2] = hi.x;
            dst[3] = hi.y;
        }
        __syncthreads();
        SWEEP_STAMP(3);
        // ---- 3a. deflation
        if (tid == 0) flags[0] = 0;
        __syncthreads();
        if (tid < JP) {
            double g = Gs[tid * GS + tid];
            if constexpr (CPLX) g = fmax(g, Gs[(tid ^ 1) * GS + (tid ^ 1)]); // (a complex row is null or not as a whole)
            const int z = (g > 0.0 && g <= mt.thr2) ? 1 : 0;
            zrow[tid] = z;
            perm[tid] = tid;
            if (z) flags[0] = 1;
        }
        __syncthreads();
        const bool any_null = flags[0] != 0;
        if (any_null) {
            for (int e = tid; e < JP * JP; e += NT) {
                const int i = e / JP, j = e % JP;
                if (zrow[i] || zrow[j]) Gs[i * GS + j] = 0.0;
            }
            __syncthreads();
        }
        // ---- 3b. convergence measure
        const double off = gram_offmax(Gs, red, tid);
        if (part == 0 && tid == 0) atomicMax(offmax_bits + (size_t)sw * sc.n_mats + mt.mat, (unsigned long long)__double_as_longlong(off));
        const bool skip = off <= mt.tol && !any_null; // pair already orthogonal: rows stay as they are
        if (!skip) {
            double* Ga = Gs;
            double* Gb = G2;
            double* Vc = Va;
            double* Vn = Vb;
            bool pos_is_interleaved = false; // cross-only solve: position 2k holds row k of block P, 2k + 1 row k of block Q
            // (cross_every < 0: adaptive -- at least four full rounds per sweep of a matrix, whatever its block count)
            const int cross_k = sc.cross_every > 0 ? sc.cross_every : max(4, (mt.nb + 2) / 4);
            const bool cross = sc.cross_every != 0 && !any_null && mt.nb >= sc.cross_min_nb && (round % cross_k) != 0; // (the same in every part)
            if constexpr (CPLX) {
                // ---- 3c'. rows are the interleaved embedding of complex rows: structure-preserving pivot solve by wave 0
                //           (work arrays in G2, rotations in Vb, result M(Q_c) in Va, eigenvalues on the diagonal of Gs)
                if (tid < 64) hermitian_pivot_solve(Gs, G2, Vb, Va, Gs, max_inner, tid, 0.25 * mt.tol * mt.tol, cross);
                __syncthreads();
            } else {
            // ---- 3c. eigensolve in position space (see jacobi_round_kernel)
            auto cpos = [](int i) { return i < JB ? 2 * i : 2 * (i - JB) + 1; };     // index -> interleaved position
            pos_is_interleaved = cross;
            if (cross) {
                for (int e = tid; e < JP * JP; e += NT) {
                    const int i = e / JP, j = e % JP;
                    Gb[cpos(i) * GS + cpos(j)] = Ga[i * GS + j];
                }
                for (int e = tid; e < JP * VS; e += NT) Va[e] = (cpos(e / VS) == (e % VS)) ? 1.0 : 0.0;
                double* t = Ga;
                Ga = Gb;
                Gb = t;
            } else
                for (int e = tid; e < JP * VS; e += NT) Va[e] = ((e / VS) == (e % VS)) ? 1.0 : 0.0;
            __syncthreads();
            {
                const int pr = tid >> 4, pc = tid & 15;
                const int r0 = 2 * pr, c0 = 2 * pc;
                // destinations of a step: the ring of all 32 positions, or (cross) even positions stay and odd ones move on by one pair
                const int dr0 = cross ? r0 : ring_next(r0), dr1 = cross ? ((r0 + 3) & (JP - 1)) : ring_next(r0 + 1);
                const int dc0 = cross ? c0 : ring_next(c0), dc1 = cross ? ((c0 + 3) & (JP - 1)) : ring_next(c0 + 1);
                const int n_steps = cross ? JB : JP - 1;
                for (int sweep = 0; sweep < (off <= mt.tol ? 0 : max_inner); ++sweep) {
                    for (int r = 0; r < n_steps; ++r) {
                        const d2 g0 = *reinterpret_cast<const d2*>(Ga + r0 * GS + c0);
                        const d2 g1 = *reinterpret_cast<const d2*>(Ga + (r0 + 1) * GS + c0);
                        const d2 ac = *reinterpret_cast<const d2*>(Ga + c0 * GS + c0);
                        const double dc = Ga[(c0 + 1) * GS + c0 + 1];
                        const d2 v0 = *reinterpret_cast<const d2*>(Vc + r0 * VS + c0);
                        const d2 v1 = *reinterpret_cast<const d2*>(Vc + (r0 + 1) * VS + c0);
                        // ONE rotation chain per thread (its column pair; the eigensolver is bound by the f64 instruction count of
                        // its single wave per SIMD: scripts/probes/eig_step_probe.hip, 1196 -> 935 cycles per step); the row
                        // pair's rotation is the one lane pc == pr of the same 16-lane row has just derived
                        double c1, s1, c2, s2;
                        jacobi_rot_bf(ac.x, dc, ac.y, c2, s2, 0.25 * mt.tol * mt.tol);
                        const int rsrc = (lane & 48) | pr;
                        c1 = __shfl(c2, rsrc);
                        s1 = __shfl(s2, rsrc);
                        const double hik = c1 * g0.x - s1 * g1.x, hil = c1 * g0.y - s1 * g1.y;
                        const double hjk = s1 * g0.x + c1 * g1.x, hjl = s1 * g0.y + c1 * g1.y;
                        double nik = c2 * hik - s2 * hil, nil = s2 * hik + c2 * hil;
                        double njk = c2 * hjk - s2 * hjl, njl = s2 * hjk + c2 * hjl;
                        if (pr == pc) {
                            nil = 0.0;
                            njk = 0.0;
                        }
                        Gb[dr0 * GS + dc0] = nik;
                        Gb[dr0 * GS + dc1] = nil;
                        Gb[dr1 * GS + dc0] = njk;
                        Gb[dr1 * GS + dc1] = njl;
                        Vn[r0 * VS + dc0] = c2 * v0.x - s2 * v0.y;
                        Vn[r0 * VS + dc1] = s2 * v0.x + c2 * v0.y;
                        Vn[(r0 + 1) * VS + dc0] = c2 * v1.x - s2 * v1.y;
                        Vn[(r0 + 1) * VS + dc1] = s2 * v1.x + c2 * v1.y;
                        __syncthreads();
                        double* t = Ga;
                        Ga = Gb;
                        Gb = t;
                        t = Vc;
                        Vc = Vn;
                        Vn = t;
                    }
                    if (sweep + 1 < max_inner) {
                        const double off_in = gram_offmax(Ga, red, tid);
                        if (off_in <= 0.25 * mt.tol) break;
                    }
                }
            }
            } // (real pivot solve)
            __syncthreads();
            SWEEP_STAMP(4);
            if (off > mt.tol && tid < 64) {
                // rank of every row by descending norm (padding rows last): lane i of wave 0 holds key i and takes the other
                // keys from the lanes by v_readlane (a scalar broadcast) instead of 32 dependent LDS reads: 1.6 -> 0.4 us
                const int li = lane & (JP - 1);
                const int li_idx = pos_is_interleaved ? ((li & 1) ? JB + (li >> 1) : (li >> 1)) : li; // the row that sits at position li
                const double g = xrow(li_idx, P, Q) < mt.nv ? Ga[li * GS + li] : -1.0e300;
                const int glo = __double2loint(g), ghi = __double2hiint(g);
                int rk = 0;
#pragma unroll
                for (int j = 0; j < JP; ++j) {
                    const double gj = __hiloint2double(__builtin_amdgcn_readlane(ghi, j), __builtin_amdgcn_readlane(glo, j));
                    rk += (gj > g || (gj == g && j < li)) ? 1 : 0;
                }
                if (lane < JP) perm[rk] = lane;
            }
            __syncthreads();
            // ---- 4. X <- Qm^T X on the own chunks, stored write-through for the next owner
            for (int e = tid; e < JP * JP; e += NT) {
                const int k = e / JP, m = e % JP;
                Qs[k * QS + m] = Vc[k * VS + perm[m]];
            }
            if (tid < JP) { // (perm holds positions; zrow is per row of the pair)
                const int pp = perm[tid];
                zout[tid] = zrow[pos_is_interleaved ? ((pp & 1) ? JB + (pp >> 1) : (pp >> 1)) : pp];
            }
            __syncthreads();
            SWEEP_STAMP(5);
            if (ct > nW) { // J is updated in this round: its rows must have been through every earlier round's update
                if (tid == 0) {
                    bool ok = spin_until(readyJ + (size_t)P * G + part, (unsigned int)ground, sc.err);
                    ok = ok && spin_until(readyJ + (size_t)Q * G + part, (unsigned int)ground, sc.err);
                    flags[1] = ok ? 1 : 0;
                }
                __syncthreads();
                if (!flags[1]) return;
#pragma unroll
                for (int c = 0; c < SW_PRE; ++c) {
                    if (c >= pre_limit && c < ct) {
#pragma unroll
                        for (int kk = 0; kk < JP / 4; ++kk) xpre[c][kk] = chunk_load(c, 4 * kk + ukq);
                    }
                }
            }
            if (ct > 0) {
                const double* ap0 = Qs + (lane >> 4) * QS + (lane & 15);
                const double* ap1 = ap0 + 16;
                double a0[JP / 4], a1[JP / 4];
#pragma unroll
                for (int kk = 0; kk < JP / 4; ++kk) {
                    a0[kk] = ap0[kk * 4 * QS];
                    a1[kk] = ap1[kk * 4 * QS];
                }
                int zr[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) zr[i][q] = zout[i * 16 + (lane >> 4) + 4 * q];
                auto apply = [&](int c, const double (&xb)[JP / 4]) {
                    d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
                    for (int kk = 0; kk < JP / 4; ++kk) {
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], xb[kk], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], xb[kk], acc[1], 0, 0, 0);
                    }
                    const bool zero_null = c < nW;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int row = i * 16 + (lane >> 4) + 4 * q;
                            bool in_w;
                            const unsigned int o = chunk_off(c, row, in_w);
                            Q8 v;
                            v.d = (zero_null && zr[i][q]) ? 0.0 : acc[i][q];
                            if (in_w) __builtin_amdgcn_raw_buffer_store_b64(v.u, rsW, o, 0, 16);
                            else __builtin_amdgcn_raw_buffer_store_b64(v.u, rsJ, o, 0, 16);
                        }
                };
#pragma unroll
                for (int c = 0; c < SW_PRE; ++c)
                    if (c < ct) apply(c, xpre[c]);
                if (ct > SW_PRE) {
                    double xb[JP / 4], xn[JP / 4];
#pragma unroll
                    for (int kk = 0; kk < JP / 4; ++kk) xb[kk] = chunk_load(SW_PRE, 4 * kk + ukq);
                    for (int c = SW_PRE; c < ct; ++c) {
                        if (c + 1 < ct) {
#pragma unroll
                            for (int kk = 0; kk < JP / 4; ++kk) xn[kk] = chunk_load(c + 1, 4 * kk + ukq);
                        }
                        apply(c, xb);
#pragma unroll
                        for (int kk = 0; kk < JP / 4; ++kk) xb[kk] = xn[kk];
                    }
                }
            }
        }
        if (defer && !skip) { // the J half of this update waits for the next round's exchange
            pendP = P;
            pendQ = Q;
            pend_ground = ground;
        }
        // ---- 5. hand the two row blocks on: every storing wave drains, barrier, ONE lane publishes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (skip && nJ > 0) { // (J untouched, but it may only be passed on once the earlier rounds' updates are in)
                (void)spin_until(readyJ + (size_t)P * G + part, (unsigned int)ground, sc.err);
                (void)spin_until(readyJ + (size_t)Q * G + part, (unsigned int)ground, sc.err);
            }
            __hip_atomic_store(ready + (size_t)P * G + part, (unsigned int)(ground + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ready + (size_t)Q * G + part, (unsigned int)(ground + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(defer && !skip)) { // (J is up to date: updated above, or the pair was skipped)
                __hip_atomic_store(readyJ + (size_t)P * G + part, (unsigned int)(ground + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(readyJ + (size_t)Q * G + part, (unsigned int)(ground + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // one-launch form: this entry's share of the sweep is done (its off-norm contributions -- atomics of this very
            // lane, completed: they return a value the drain above waited for -- are in)
            if (n_sw > 1 && round == mt.nb - 2) __hip_atomic_fetch_add(sc.arrive + mt.mat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SWEEP_STAMP(6);
    }
    } // entries x rounds
    if (n_sw == 1) break;
    // ---- end of a sweep, one-launch form: the host's convergence test (jacobi_orthogonalise), on the device.  Every entry
    //      of a matrix sees the same off-norms, hence takes the same decision; an entry waits until all entries of its
    //      matrix have finished the sweep, workgroups whose entries have all converged leave.
    __syncthreads();
    if (tid == 0) {
        int all = 1, ok = 1;
        for (int ent = went.x; ent < went.y && ok; ++ent) {
            if (edone[ent - went.x]) continue;
            const int2 wm = sc.wgmap[ent];
            const RPair mt = pairs[wm.x];
            const unsigned int entries = (unsigned int)(mt.nb / 2) * (unsigned int)mt.pad[0];
            if (!spin_until<32>(sc.arrive + mt.mat, entries * (unsigned int)(sw + 1), sc.err)) { // (every workgroup of the matrix polls this word)
                ok = 0;
                break;
            }
            const double off = __longlong_as_double((long long)__hip_atomic_load(offmax_bits + (size_t)sw * sc.n_mats + mt.mat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const double prev = sw ? __longlong_as_double((long long)__hip_atomic_load(offmax_bits + (size_t)(sw - 1) * sc.n_mats + mt.mat,
                                                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                   : 1e300;
            const bool stagnated = sw + 1 >= 6 && off <= 64.0 * mt.tol && off >= 0.5 * prev;
            const bool predicted = off <= 0.1 * sqrt(mt.tol) && prev < 1.0 && off <= prev * sqrt(prev) && off <= kPredictQuad * prev * prev;
            if (off <= mt.tol || stagnated || predicted) {
                edone[ent - went.x] = 1;
                if (mt.slot == 0 && wm.y == 0) sc.nsw[mt.mat] = sw + 1;
            } else all = 0;
        }
        flags[1] = ok;
        flags[2] = all;
    }
    __syncthreads();
    if (!flags[1] || flags[2]) return;
    } // sweeps
#undef SWEEP_STAMP
}

} // namespace

int jacobi_orthogonalise(cyb_ctx_t ctx, const std::vector<JMat>& h_mats, int max_sweeps,
                         std::vector<int32_t>& sweeps_out, bool cplx)
{   // cplx: the rows are the interleaved real embedding of complex rows (structure-preserving pivot solve; persistent path only)
    const int n = (int)h_mats.size();
    sweeps_out.assign((size_t)n, 0);
    if (n == 0) return CYB_OK;
    hipStream_t st = ctx->stream;

    std::vector<int> active;
    for (int i = 0; i < n; ++i) {
        if (h_mats[(size_t)i].nv <= 1) sweeps_out[(size_t)i] = 0; // nothing to orthogonalise
        else active.push_back(i);
    }
    // convergence words of a sweep: [off-norm bits per matrix | error word], the head of the zeroed scratch block, so
    // that a sweep costs ONE memset before and ONE read-back after its kernel(s)
    // One launch for ALL sweeps (persistent sweep kernel, convergence test on the device): saves
    // the read-back, the host's turn-around and the launch of every sweep but the first (40-100 us each).  Correct (same
    // sweep counts, tests green) but NOT faster for large matrices: the chi=4096 list 34.0 against 34.2-34.4 ms per batched SVD, the toy DMRG
    // at chi=256 0.27-0.29 against 0.28-0.31 s per sweep (both inside the run-to-run spread), and a single rank-deficient 1442^2
    // block 29.9 against 28.9 ms -- its one kernel runs 15.7 ms where the ten per-sweep kernels sum to 14.5 ms (the sweep-end
    // wait of 192 workgroups on one word costs more than the launch boundary it replaces).  The control words of this form
    // are [off-norm bits per (sweep, matrix) | arrivals per matrix | sweeps per matrix].
    // Default therefore: lists of SMALL matrices only (at most 16 row blocks each -- the sectors of a DMRG bond around chi = 256:
    // few workgroups per matrix wait at a sweep's end, and the seven read-backs are a visible part of a 2 ms call; toy DMRG,
    // three runs each: 0.271 / 0.275 / 0.280 against 0.282 / 0.283 / 0.285 s per sweep).  CYB_JACOBI_ONELAUNCH=1: every list,
    // CYB_JACOBI_PERSWEEP=1: none.
    static const bool one_launch_ok = getenv("CYB_JACOBI_PERSWEEP") == nullptr && getenv("CYB_JACOBI_TRACE") == nullptr &&
                                      getenv("CYB_JACOBI_STAMPS") == nullptr && getenv("CYB_JACOBI_NOPREDICT") == nullptr;
    static const bool one_launch_all = getenv("CYB_JACOBI_ONELAUNCH") != nullptr;
    int nb_max_all = 0;
    for (int i = 0; i < n; ++i) nb_max_all = std::max(nb_max_all, h_mats[(size_t)i].nb);
    const int n_sw_dev = (one_launch_ok && (one_launch_all || nb_max_all <= 16)) ? max_sweeps : 1;
    const size_t w_off = (size_t)n * (size_t)n_sw_dev; // off-norm words
    const size_t b_off = (sizeof(unsigned long long) * w_off + 2 * sizeof(unsigned int) * (size_t)n + 15) / 16 * 16;
    std::vector<unsigned long long> h_off(b_off / 8 + 2);
    bool one_launch_done = false;
    // descriptors of the persistent sweep stay on the device while the set of active matrices does not change
    std::vector<int> cached_order, cached_G;
    uint64_t cached_at = 0;
    void* cached_img = nullptr;
    size_t cached_map_off = 0;
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
        const JMat* d_mats = nullptr; // (uploaded by the launch-per-round paths only)
        void* d_wl = nullptr;
        // Two ways to run a round.  FUSED (jacobi_round_kernel): one launch, the parts of a pair exchange their Gram
        // partials inside it -- the latency path, for rounds with few pairs (the large blocks once the small ones have
        // converged, a single block, a DMRG-sized list).  SPLIT (jacobi_gram_kernel + jacobi_update_kernel): two launches,
        // no residency constraint, the update spread over three workgroups per CU -- the throughput path for rounds
        // with many pairs, where the f64 MFMA rate of the whole chip (not a dependency chain) bounds the update.
        // Within a sweep the pair count only falls (matrices drop out of the lockstep as r passes their nb - 1), so a
        // sweep starts split and turns fused.
        static const bool legacy = getenv("CYB_JACOBI_LEGACY") != nullptr;
        static const int fused_max = getenv("CYB_JACOBI_FUSED_MAX") ? atoi(getenv("CYB_JACOBI_FUSED_MAX")) : 64;
        // inner sweeps: few while far from convergence (the outer iteration repeats anyway)
        // inner Jacobi sweeps per pair visit: measurements and a numpy model agree that more than two
        // buy no outer sweeps, and one is fastest overall (measured: 81 vs 88 ms on the chi=4096 list)
        // ... except for batches of small matrices (at most 8 row blocks each: the DMRG regime around chi = 256), where a
        // round is all latency and an outer sweep costs a host read-back: three inner sweeps save outer sweeps there
        // (toy DMRG chi=256, eleven sweeps: 6.6-6.7 -> 6.3-6.6 s; the 13-block chi=1024 list, 23 row blocks: 11.6 -> 14.9 ms, so not there)
        static const int inner_env = getenv("CYB_JACOBI_INNER") ? atoi(getenv("CYB_JACOBI_INNER")) : 0;
        const int max_inner = inner_env > 0 ? inner_env : (max_nb <= 8 ? 3 : 1);
        constexpr int kGmax = RGMAX;
        const size_t np = wl.size();
        // ---- scratch of both paths in one grow-only workspace: [tickets | error word | Gram partials | split-path hand-off]
        const size_t b_cnt = (sizeof(unsigned int) * np * (size_t)std::max(max_nb - 1, 1) + 15) / 16 * 16;
        const size_t b_gpart = sizeof(double) * np * kGmax * JP * JP, b_q = sizeof(double) * np * JP * JP;
        const size_t b_z = sizeof(int32_t) * np * JP, b_f = (sizeof(int32_t) * np + 15) / 16 * 16, b_c = (sizeof(unsigned int) * np + 15) / 16 * 16;
        size_t n_ready = 0; // persistent sweep: one counter per (matrix, block, part), at most RGMAX parts
        for (int m : order) n_ready += (size_t)h_mats[(size_t)m].nb * RGMAX;
        const size_t b_ready = (sizeof(unsigned int) * (2 * n_ready + np) + 15) / 16 * 16; // ready counters (W and J) + one ticket per pair
        void* wsp = nullptr;
        status = ctx->workspace(b_off + 16 + b_cnt + b_ready + 2 * b_gpart + 2 * (b_q + b_z + b_f) + b_c + 1024, &wsp, 2);
        if (status != CYB_OK) break;
        char* bp = static_cast<char*>(wsp);
        RScratch rs;
        // one zeroed block: [off-norm words | error word | tickets of the round kernel | tickets + ready counters of the sweep kernel]
        unsigned long long* d_off = reinterpret_cast<unsigned long long*>(bp);
        rs.err = reinterpret_cast<unsigned int*>(bp + b_off);
        rs.cnt = reinterpret_cast<unsigned int*>(bp + b_off + 16);
        unsigned int* d_ready = reinterpret_cast<unsigned int*>(bp + b_off + 16 + b_cnt);
        bp += b_off + 16 + b_cnt + b_ready;
        rs.gpart = reinterpret_cast<double*>(bp);
        bp += 2 * b_gpart;
        rs.np = (int)np;
        rs.stamps = nullptr;
        JScratch sc;
        sc.gpart = rs.gpart; // (a round runs one path or the other)
        for (int h = 0; h < 2; ++h) {
            sc.qout[h] = reinterpret_cast<double*>(bp);
            bp += b_q;
        }
        for (int h = 0; h < 2; ++h) {
            sc.zout[h] = reinterpret_cast<int32_t*>(bp);
            bp += b_z;
        }
        for (int h = 0; h < 2; ++h) {
            sc.flag[h] = reinterpret_cast<int32_t*>(bp);
            bp += b_f;
        }
        sc.cnt = reinterpret_cast<unsigned int*>(bp);
        if (hipMemsetAsync(wsp, 0, b_off + 16 + b_cnt + b_ready, st) != hipSuccess) {
            status = CYB_ERR_HIP;
            break;
        }
        std::vector<RPair> rp(np);
        for (size_t k = 0; k < np; ++k) {
            const JMat& jm = h_mats[(size_t)wl[k].mat];
            RPair& q = rp[k];
            q.W = jm.W;
            q.J = jm.J;
            q.tol = jm.tol;
            q.thr2 = jm.thr2;
            q.nvp = jm.nvp;
            q.lenp = jm.lenp;
            q.nb = jm.nb;
            q.nv = jm.nv;
            q.mat = wl[k].mat;
            q.slot = wl[k].slot;
            q.pad[0] = q.pad[1] = 0;
        }
        void* d_rp = nullptr; // (uploaded by the launch-per-round paths only)
        static const bool want_stamps = getenv("CYB_JACOBI_STAMPS") != nullptr;
        static unsigned long long* d_stamps = nullptr; // diagnostic runs only: 2048 workgroups x 8 stamps
        if (want_stamps) {
            if (!d_stamps && hipMalloc(reinterpret_cast<void**>(&d_stamps), sizeof(unsigned long long) * 8 * 2048) != hipSuccess) d_stamps = nullptr;
            rs.stamps = d_stamps;
        }
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_round_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)ROUND_LDS_BYTES) != hipSuccess) {
                set_error("jacobi: cannot reserve %zu bytes of LDS for the round kernel", ROUND_LDS_BYTES);
                status = CYB_ERR_HIP;
                break;
            }
            attr_set = true;
        }
        bool any_j = false;
        for (int m : order) any_j = any_j || h_mats[(size_t)m].J != nullptr;
        static const bool defer_j = getenv("CYB_JACOBI_NODEFER") == nullptr;
        static const int g_env = getenv("CYB_JACOBI_G") ? atoi(getenv("CYB_JACOBI_G")) : 0;
        static const int u_max = getenv("CYB_JACOBI_UMAX") ? atoi(getenv("CYB_JACOBI_UMAX")) : 16;
        // ---- persistent sweep (jacobi_sweep_kernel): every pair of every matrix gets its workgroups for the whole sweep;
        //      possible when they all fit on the chip at once (one workgroup per CU)
        static const bool no_sweep = getenv("CYB_JACOBI_NOSWEEP") != nullptr;
        bool did_sweep = false;
        bool big_ok = true;
        for (int m : order)
            big_ok = big_ok && (size_t)h_mats[(size_t)m].nvp * (size_t)std::max(h_mats[(size_t)m].lenp, h_mats[(size_t)m].nvp) * 8 < ((size_t)1 << 31);
        const size_t sweep_slots = (size_t)SWEEP_WG_PER_CU * (size_t)ctx->n_cu; // resident workgroups the sweep kernel may use
        // (lists with more pairs than resident workgroups: short matrices share workgroups -- CYB_JACOBI_NOSHARE sends them
        //  to the launch-per-round paths as before)
        static const bool no_share = getenv("CYB_JACOBI_NOSHARE") != nullptr;
        if (!legacy && !no_sweep && big_ok && (np <= sweep_slots || !no_share)) {
            // parts per pair, matrix by matrix: start with one, then keep giving a CU per pair to the matrix whose sweep
            // is the longest chain (rounds x per-round work / parts) while the chip has room
            std::vector<int> Gm((size_t)n, 1);
            auto chain = [&](int m) {
                const JMat& jm = h_mats[(size_t)m];
                const double per_round = 8.0 + 22.0 * (double)(jm.lenp + (jm.J ? jm.nvp : 0)) / 2240.0 / (double)Gm[(size_t)m]
                                         + (Gm[(size_t)m] > 1 ? 4.0 : 0.0); // us: fixed part + MFMA share (+ exchange)
                return per_round * (double)(jm.nb - 1);
            };
            size_t used = np;
            while (true) {
                int best = -1;
                double worst = 0.0;
                for (int m : order) {
                    const JMat& jm = h_mats[(size_t)m];
                    const int cw_ = jm.lenp / 64;
                    if (Gm[(size_t)m] >= RGMAX || Gm[(size_t)m] >= cw_ || used + (size_t)jm.nb / 2 > sweep_slots) continue;
                    const double c = chain(m);
                    if (c > worst) {
                        worst = c;
                        best = m;
                    }
                }
                if (best < 0) break;
                ++Gm[(size_t)best];
                used += (size_t)h_mats[(size_t)best].nb / 2;
            }
            if (g_env > 0)
                for (int m : order) Gm[(size_t)m] = std::max(1, std::min({g_env, RGMAX, h_mats[(size_t)m].lenp / 64}));
            size_t total = 0;
            for (int m : order) total += (size_t)h_mats[(size_t)m].nb / 2 * (size_t)Gm[(size_t)m];
            if (total <= sweep_slots || (!no_share && total == np)) {
                std::vector<int> flag_base((size_t)n, 0);
                int fb = (int)np; // the tickets come first in the zeroed block
                for (int m : order) {
                    flag_base[(size_t)m] = fb;
                    fb += h_mats[(size_t)m].nb * Gm[(size_t)m];
                }
                std::vector<int2> wgmap;
                for (size_t k = 0; k < np; ++k) {
                    rp[k].pad[0] = Gm[(size_t)wl[k].mat];
                    rp[k].pad[1] = flag_base[(size_t)wl[k].mat];
                    for (int g = 0; g < Gm[(size_t)wl[k].mat]; ++g) wgmap.push_back(make_int2((int)k, g));
                }
                // More entries than resident workgroups (a list of many large matrices: 23 hermitian blocks up to 1238^2 have
                // 301 pairs): workgroups take several entries and run them one after the other in every round.  A matrix
                // whose pairs sit on shared workgroups advances at 1/m of the pace, so the longest chains keep workgroups of
                // their own and the short matrices share: split the list (sorted by row blocks) where
                // max(rounds of the longest, m x rounds of the longest shared) is smallest.
                std::vector<int2> wgent;
                if (wgmap.size() > sweep_slots) {
                    std::vector<size_t> ent_before; // entries of the matrices before position i of `order`
                    size_t acc = 0;
                    for (int m : order) {
                        ent_before.push_back(acc);
                        acc += (size_t)h_mats[(size_t)m].nb / 2;
                    }
                    ent_before.push_back(acc);
                    size_t best_i = 0, best_m = 0;
                    double best_cost = 1e300;
                    for (size_t i = 0; i <= order.size(); ++i) {
                        const size_t D = ent_before[i], S = acc - D;
                        if (D > sweep_slots || (S > 0 && D >= sweep_slots)) break;
                        const size_t m = S ? (S + (sweep_slots - D) - 1) / (sweep_slots - D) : 1;
                        const double own = (double)(h_mats[(size_t)order[0]].nb - 1);
                        const double shared = i < order.size() ? (double)m * (double)(h_mats[(size_t)order[i]].nb - 1) : 0.0;
                        const double cost = std::max(own, shared);
                        if (cost < best_cost) {
                            best_cost = cost;
                            best_i = i;
                            best_m = m;
                        }
                    }
                    const size_t D = ent_before[best_i];
                    for (size_t e = 0; e < D; ++e) wgent.push_back(make_int2((int)e, (int)e + 1));
                    for (size_t e = D; e < wgmap.size(); e += best_m)
                        wgent.push_back(make_int2((int)e, (int)std::min(e + best_m, wgmap.size())));
                }
                const size_t n_wg = wgent.empty() ? wgmap.size() : wgent.size();
                // ONE upload ([pair descriptors | entry map | workgroup -> entries]), and none at all while the active set and
                // its parts are those of the previous sweep and the ring slot is still alive
                std::vector<int> g_now;
                for (int m : order) g_now.push_back(Gm[(size_t)m]);
                const size_t map_off = (sizeof(RPair) * np + 255) / 256 * 256;
                const size_t ent_off = (map_off + sizeof(int2) * wgmap.size() + 255) / 256 * 256;
                if (!(cached_img && order == cached_order && g_now == cached_G && ctx->n_uploads - cached_at < (uint64_t)cyb_ctx_s::kSlots / 2 - 1)) {
                    std::vector<char> img(ent_off + sizeof(int2) * wgent.size());
                    memcpy(img.data(), rp.data(), sizeof(RPair) * np);
                    memcpy(img.data() + map_off, wgmap.data(), sizeof(int2) * wgmap.size());
                    if (!wgent.empty()) memcpy(img.data() + ent_off, wgent.data(), sizeof(int2) * wgent.size());
                    status = ctx->upload(img.data(), img.size(), &cached_img);
                    if (status != CYB_OK) break;
                    cached_order = order;
                    cached_G = g_now;
                    cached_at = ctx->n_uploads;
                    cached_map_off = map_off;
                }
                void* d_rp2 = cached_img;
                void* d_map = static_cast<char*>(cached_img) + cached_map_off;
                static bool attr2_set = false;
                if (!attr2_set) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_sweep_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)SWEEP_LDS_BYTES) != hipSuccess ||
                        hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_sweep_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)SWEEP_LDS_BYTES) != hipSuccess) {
                        set_error("jacobi: cannot reserve %zu bytes of LDS for the sweep kernel", SWEEP_LDS_BYTES);
                        status = CYB_ERR_HIP;
                        break;
                    }
                    attr2_set = true;
                }
                SScratch ss;
                ss.gpart = rs.gpart;
                ss.ticket = d_ready;           // [0, np): tickets; behind them the ready counters (flag_base offsets)
                ss.ready = d_ready;
                ss.readyJ = d_ready + n_ready; // (same offsets, second array)
                static const bool no_defer_j = getenv("CYB_JACOBI_NODEFERJ") != nullptr;
                ss.defer_j = no_defer_j ? 0 : 1;
                static const int cross_env = getenv("CYB_JACOBI_CROSS") ? atoi(getenv("CYB_JACOBI_CROSS")) : -1;
                // (lists of small matrices run three inner sweeps per pivot solve: cross-only there is opt-in, CYB_JACOBI_CROSS_SMALL=1)
                static const bool cross_small = getenv("CYB_JACOBI_CROSS_SMALL") != nullptr;
                ss.cross_every = (max_inner == 1 || cross_small) ? cross_env : 0;
                static const int cross_min_nb = getenv("CYB_JACOBI_CROSS_MINNB") ? atoi(getenv("CYB_JACOBI_CROSS_MINNB")) : 0;
                ss.cross_min_nb = cross_min_nb;
                ss.err = rs.err;
                ss.wgmap = static_cast<const int2*>(d_map);
                ss.wgent = wgent.empty() ? nullptr : reinterpret_cast<const int2*>(static_cast<char*>(cached_img) + ent_off);
                ss.stamps = wgent.empty() ? rs.stamps : nullptr;
                ss.stamp_round = std::min(3, max_nb - 2);
                size_t max_ent = 1;
                for (const int2& we : wgent) max_ent = std::max(max_ent, (size_t)(we.y - we.x));
                const bool one_launch = n_sw_dev > 1 && sweep == 1 && max_ent <= (size_t)SW_MAX_ENT;
                ss.n_sweeps = one_launch ? n_sw_dev : 1;
                ss.n_mats = n;
                ss.arrive = reinterpret_cast<unsigned int*>(d_off + w_off);
                ss.nsw = reinterpret_cast<int*>(ss.arrive + n);
                one_launch_done = one_launch;
                if (cplx)
                    hipLaunchKernelGGL(jacobi_sweep_kernel<true>, dim3((unsigned)n_wg), dim3(NT), SWEEP_LDS_BYTES, st,
                                       static_cast<const RPair*>(d_rp2), max_inner, d_off, ss);
                else
                    hipLaunchKernelGGL(jacobi_sweep_kernel<false>, dim3((unsigned)n_wg), dim3(NT), SWEEP_LDS_BYTES, st,
                                       static_cast<const RPair*>(d_rp2), max_inner, d_off, ss);
                if (ss.stamps && wgmap.size() <= 2048) {
                    std::vector<unsigned long long> hs(8 * wgmap.size());
                    if (hipMemcpyAsync(hs.data(), rs.stamps, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost, st) == hipSuccess &&
                        hipStreamSynchronize(st) == hipSuccess) {
                        // phase times of round `stamp_round` for the workgroups of the matrix with the most blocks
                        double acc[6] = {0, 0, 0, 0, 0, 0};
                        int cntw = 0;
                        const int mbig = order.front();
                        for (size_t b = 0; b < wgmap.size(); ++b) {
                            if (wl[(size_t)wgmap[b].x].mat != mbig) continue;
                            for (int k = 0; k < 6; ++k) acc[k] += (double)(hs[8 * b + k + 1] - hs[8 * b + k]) * 0.01;
                            ++cntw;
                        }
                        if (cntw)
                            fprintf(stderr, "[jacobi sweep stamps] sweep %d round %d, largest matrix (G=%d, %d workgroups of %zu): wait %.2f load+gram %.2f "
                                            "exch %.2f defl+eig %.2f sort %.2f upd+publish %.2f us\n",
                                    sweep, ss.stamp_round, Gm[(size_t)mbig], cntw, wgmap.size(), acc[0] / cntw, acc[1] / cntw,
                                    acc[2] / cntw, acc[3] / cntw, acc[4] / cntw, acc[5] / cntw);
                    }
                }
                did_sweep = true;
            }
        }
        if (!did_sweep && cplx) {
            set_error("block-Jacobi on embedded complex rows needs the persistent sweep kernel (list too large for one launch)");
            status = CYB_ERR_UNSUPPORTED;
            break;
        }
        if (!did_sweep) {
        cached_img = nullptr; // (the uploads below recycle the ring)
        {
            void* d_mats_v = nullptr;
            status = ctx->upload(h_mats.data(), sizeof(JMat) * (size_t)n, &d_mats_v);
            if (status == CYB_OK) status = ctx->upload(wl.data(), sizeof(JWork) * wl.size(), &d_wl);
            if (status == CYB_OK) status = ctx->upload(rp.data(), sizeof(RPair) * np, &d_rp);
            if (status != CYB_OK) break;
            d_mats = static_cast<const JMat*>(d_mats_v);
            if (hipMemsetAsync(sc.cnt, 0, b_c, st) != hipSuccess) {
                status = CYB_ERR_HIP;
                break;
            }
        }
        int pend_round = -1;   // split path: round whose J half is still to be applied ...
        size_t pend_cnt = 0;   // ... for this many pairs
        auto flush_pending = [&]() {
            if (pend_round < 0) return;
            const int U = (int)std::min<size_t>(16, std::max<size_t>(1, (size_t)(3 * ctx->n_cu) / pend_cnt));
            hipLaunchKernelGGL(jacobi_update_kernel, dim3((unsigned)(pend_cnt * U)), dim3(NT), 0, st, d_mats,
                               static_cast<const JWork*>(d_wl), pend_round, U, sc, pend_round & 1, 0, 1);
            pend_round = -1;
        };
        for (int r = 0; r < max_nb - 1; ++r) {
            size_t cnt = 0; // grid = prefix of the work list holding matrices with nb - 1 > r
            for (int m : order) {
                if (h_mats[(size_t)m].nb - 1 > r) cnt += (size_t)h_mats[(size_t)m].nb / 2;
                else break;
            }
            if (cnt == 0) break;
            const bool fused = !legacy && cnt <= (size_t)fused_max && cnt <= (size_t)ctx->n_cu;
            if (fused) {
                flush_pending(); // (the J half of the last split round)
                // The G parts of a pair wait for one another: all of them must be resident, i.e. at most one workgroup
                // per CU (the kernel's LDS request allows no second one).
                int G = (int)std::min<size_t>(kGmax, (size_t)ctx->n_cu / cnt);
                if (g_env > 0 && (size_t)g_env * cnt <= (size_t)ctx->n_cu) G = std::min(g_env, kGmax);
                G = std::max(G, 1);
                hipLaunchKernelGGL(jacobi_round_kernel, dim3((unsigned)(cnt * G)), dim3(NT), ROUND_LDS_BYTES, st,
                                   static_cast<const RPair*>(d_rp), r, G, max_inner, d_off, rs);
                if (rs.stamps && r == max_nb / 2 && cnt * (size_t)G <= 2048) { // diagnostic: phase times of one round, all workgroups
                    std::vector<unsigned long long> hs(8 * cnt * G);
                    if (hipMemcpyAsync(hs.data(), rs.stamps, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost, st) == hipSuccess &&
                        hipStreamSynchronize(st) == hipSuccess) {
                        double acc[6] = {0, 0, 0, 0, 0, 0}, tot = 0, mx = 0;
                        for (size_t b = 0; b < cnt * G; ++b) {
                            for (int k = 0; k < 6; ++k) acc[k] += (double)(hs[8 * b + k + 1] - hs[8 * b + k]) * 0.01;
                            const double t = (double)(hs[8 * b + 6] - hs[8 * b]) * 0.01;
                            tot += t;
                            mx = std::max(mx, t);
                        }
                        const double nb_ = (double)(cnt * G);
                        fprintf(stderr, "[jacobi stamps] sweep %d round %d: %zu pairs x G=%d | gram %.2f exch %.2f defl %.2f eig %.2f sort %.2f upd %.2f | wg mean %.2f max %.2f us\n",
                                sweep, r, cnt, G, acc[0] / nb_, acc[1] / nb_, acc[2] / nb_, acc[3] / nb_, acc[4] / nb_, acc[5] / nb_, tot / nb_, mx);
                    }
                }
                continue;
            }
            // ---- split round: spread every pair over enough workgroups to fill the chip (2 resident per CU for A, more for B)
            const int slots = 2 * ctx->n_cu;
            // measured (chi=4096 list): splitting the Gram further than 2 ways costs more in the partial
            // exchange than it saves (G = 1: 72.2, 2: 71.9, 4: 75.4, 8: 88.7 ms per batched SVD)
            int G = g_env > 0 ? g_env : ((size_t)2 * cnt <= (size_t)slots ? 2 : 1);
            G = std::min(G, kGmax);
            // the update kernel holds 3 workgroups per CU: one full wave of workgroups, no tail
            const int U = (int)std::min<size_t>((size_t)u_max, std::max<size_t>(1, (size_t)(3 * ctx->n_cu) / cnt));
            const int buf = r & 1;
            const int n_gram = (int)(cnt * G);
            int UJ = 1;
            size_t n_jwg = 0;
            if (pend_round >= 0) { // the J half of the previous round rides along in the idle workgroup slots
                const long long free_slots = (long long)slots - (long long)n_gram;
                UJ = (int)std::max<long long>(1, std::min<long long>(8, free_slots / (long long)pend_cnt));
                n_jwg = pend_cnt * (size_t)UJ;
            }
            hipLaunchKernelGGL(jacobi_gram_kernel, dim3((unsigned)(n_gram + n_jwg)), dim3(NT), 0, st, d_mats,
                               static_cast<const JWork*>(d_wl), r, G, max_inner, d_off, sc, buf, n_gram, pend_round, UJ);
            const bool defer = defer_j && any_j;
            hipLaunchKernelGGL(jacobi_update_kernel, dim3((unsigned)(cnt * U)), dim3(NT), 0, st, d_mats,
                               static_cast<const JWork*>(d_wl), r, U, sc, buf, 1, defer ? 0 : 1);
            pend_round = defer ? r : -1;
            pend_cnt = cnt;
        }
        flush_pending(); // the work list changes with the sweep
        } // launch-per-round paths
        if (hipGetLastError() != hipSuccess) {
            set_error("jacobi round kernels: launch failed");
            status = CYB_ERR_HIP;
            break;
        }
        unsigned int h_err = 0;
        if (ctx->d2h(h_off.data(), d_off, b_off + 16) != CYB_OK) {
            set_error("jacobi: reading the convergence flags failed: %s", hipGetErrorString(hipGetLastError()));
            status = CYB_ERR_HIP;
            break;
        }
        memcpy(&h_err, reinterpret_cast<const char*>(h_off.data()) + b_off, sizeof(unsigned int));
        if (h_err) {
            set_error("jacobi round kernel: a workgroup waited more than a second for the Gram partials of its pair "
                      "(partner workgroup not resident?)");
            status = CYB_ERR_HIP;
            break;
        }
        std::vector<int> still;
        if (one_launch_done) { // every sweep has run: matrices the device marked converged are done, the rest did not converge
            const int* h_nsw = reinterpret_cast<const int*>(reinterpret_cast<const unsigned int*>(h_off.data() + w_off) + n);
            for (int m : active) {
                if (h_nsw[m] > 0) sweeps_out[(size_t)m] = h_nsw[m];
                else still.push_back(m);
            }
            active.swap(still);
            break;
        }
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
            // `off` was measured pair by pair BEFORE this sweep rotated the pair.  In the quadratic regime the
            // sweep leaves off^2 behind: once off <= sqrt(tol)/10 the rotations just applied have already
            // taken the matrix to the rounding floor and another sweep would only re-measure it
            // (chi=4096 list: 1.8e-4 -> 2.3e-9 -> 3.4e-14 in the last three sweeps, tol 3.4e-14).
            static const bool no_predict = getenv("CYB_JACOBI_NOPREDICT") != nullptr;
            // ... but only when the decrease is actually superlinear: with clustered singular values the off-norm
            // can creep down linearly (1.7e-8 -> 4.4e-9 per sweep was measured) and a sweep is then NOT the last
            // ... and only when the last decrease WAS quadratic (off <= kPredictQuad * prev^2): a 128-fold zero eigenvalue
            // (Gram matrix of a rank-132 block, scripts/svd_fuzz.py seed 301) went 5.0e-7 -> 3.2e-10 -> 1.6e-10 -> 1.3e-14 --
            // superlinear by the test above, but the rotations inside the cluster (equal norms: large angles on couplings of
            // 1e-10) carry the couplings to the other rows from one cluster row to the next instead of annihilating them, and
            // the predicted stop left 1.6e-10 of non-orthogonality.  Quadratic steps measured here have off / prev^2 = 0.07 ... 0.2
            // (chi=4096 list: 1.8e-4 -> 2.3e-9), that one 1280.
            const double prev = prev_off[(size_t)m];
            const bool predicted = !no_predict && off <= 0.1 * std::sqrt(tol) && prev < 1.0 && off <= prev * std::sqrt(prev) &&
                                   off <= kPredictQuad * prev * prev;
            if (off <= tol || stagnated || predicted) sweeps_out[(size_t)m] = sweep;
            else still.push_back(m);
            prev_off[(size_t)m] = off;
        }
        active.swap(still);
    }
    if (status != CYB_OK) return status;
    if (!active.empty()) {
        for (int m : active) sweeps_out[(size_t)m] = -1;
        set_error("block-Jacobi did not converge within %d sweeps for %zu matrices", max_sweeps, active.size());
        return CYB_ERR_NOCONV;
    }
    return CYB_OK;
}

} // namespace cyb
