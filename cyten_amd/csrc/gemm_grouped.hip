// Grouped variable-shape fp64 block GEMM on v_mfma_f64_16x16x4_f64 (gfx950).
//
// Replaces the per-pair np.dot loop of the reference's abelian_compose_worker
// (src/backends/abelian.cpp:1424-1460; NumpyBlockBackend::matrix_dot numpy.cpp:1218-1225) with
// one launch per tile class over *all* result blocks of a tensor contraction.
//
// Kernel anatomy (one workgroup = one BMxBN tile of one result block):
//   * K runs over the concatenation of the problem's segments (the K-split pairs the reference
//     sums with Block::operator+), accumulated in MFMA accumulators -> C is written once.
//   * operands are strided views; per segment and operand the contiguous direction is either k
//     or m/n.  Global loads are 16-B vectors along the contiguous direction; the LDS image keeps
//     that direction contiguous too:  k-contiguous  -> Xs[mn][k]  (row stride BK+2 doubles),
//     mn-contiguous -> Xs[k][mn] (row stride BM+16 doubles).  Both are conflict-free for the
//     ds_read_b64 fragment reads of the 16x16x4 f64 MFMA (lane l: A[l&15][l>>4], B[l>>4][l&15]).
//   * register-staged double buffering: global loads of k-tile t+1 are in flight while the MFMAs
//     of k-tile t run; one barrier per k-tile.
//   * f64 C/D layout: col = lane&15, row = (lane>>4) + 4*reg  (NOT the f32 map,
//     cdna_hip_programming.md section 3).
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
// 16-byte vector with only 8-byte alignment guaranteed (odd leading dimensions / sliced views)
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
// Descriptor pointers are loaded from memory, so the compiler only knows them as generic (flat)
// pointers; flat loads count on lgkmcnt as well as vmcnt and serialise the LDS pipeline.  All
// operands live in HBM, so address them through the global address space explicitly.
#define GLOBAL_AS __attribute__((address_space(1)))
typedef const GLOBAL_AS double* gcptr;
typedef GLOBAL_AS double* gptr;
typedef const GLOBAL_AS d2u* gcptr2;

constexpr int BK = 16;
// s_setprio(1) around the MFMA cluster of a k-tile: the co-resident wave's staging (global loads, LDS
// writes) then yields the issue slots to the wave that is in its matrix phase.  A/B on one device:
// theta chi=4096 355 -> 339 us, uniform 4096^3 55.6 -> 57.5 TFLOP/s.
#ifndef CYB_GEMM_PRIO
#define CYB_GEMM_PRIO 1
#endif
constexpr bool kPrioMfma = CYB_GEMM_PRIO != 0;
#ifndef CYB_GEMM_FAST
#define CYB_GEMM_FAST 1
#endif
constexpr bool kFastLoop = CYB_GEMM_FAST != 0;

struct DevSeg {
    const double* A;
    const double* B;
    int64_t a_rs, a_cs, b_rs, b_cs;
    int32_t K;
    int32_t pad;
};

struct DevProb {
    double* C;
    int64_t ldc;
    int32_t M, N;
    int32_t seg_begin, seg_end;
    double alpha, beta;
    // optional left factor applied to the product before alpha/beta: C = alpha * L (A B) + beta * C, L is
    // M x M with M <= 32 (the T factor of a block reflector: saves a launch per panel step of the blocked QR)
    const double* L;
    int32_t l_rs, l_cs;
};

struct DevTile {
    int32_t prob, tm, tn, pad; // tm / tn: first row / column of the tile, pad: tile class
};

__host__ __device__ constexpr int lds_km_stride(int BMN) { return ((BMN + 16) % 32 == 16) ? BMN + 16 : BMN + 32; }
__host__ __device__ constexpr int lds_tile_doubles(int BMN)
{
    int a = BMN * (BK + 2);
    int b = BK * lds_km_stride(BMN);
    return a > b ? a : b;
}

// Load one BMN x BK operand tile into registers.  `base` points at element (mn = 0, k = 0) of the
// operand view, s_mn / s_k are its element strides, exactly one of them is 1.
// Branch-free inside a wave: out-of-range mn positions are *clamped* (their products only reach
// accumulator rows/columns that are never stored), out-of-range k positions are zeroed.  The only
// branch is the wave-uniform fast/slow choice: fast = whole 16-B vectors are in range.
template <int BMN, int NT>
__device__ __forceinline__ void load_tile(d2 (&r)[(BMN * 8 + NT - 1) / NT], const double* __restrict__ base_,
                                          int64_t s_mn, int64_t s_k, bool k_contig, int mn0, int k0,
                                          int MN, int K, int tid)
{
    constexpr int NV = (BMN * 8 + NT - 1) / NT;
    constexpr int NVEC = BMN * 8; // vectors in the tile (threads beyond them idle: NT > NVEC for the 16-wide class)
    gcptr base = (gcptr)base_;
    const bool full_k = (k0 + BK <= K);
    if (k_contig) {
        if (full_k) {
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int v = min(tid + p * NT, NVEC - 1);
                const int mn = min(mn0 + (v >> 3), MN - 1);
                const int k = k0 + 2 * (v & 7);
                r[p] = *(gcptr2)(base + (int64_t)mn * s_mn + k);
            }
        } else {
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int v = min(tid + p * NT, NVEC - 1);
                const int mn = min(mn0 + (v >> 3), MN - 1);
                const int k = k0 + 2 * (v & 7);
                gcptr row = base + (int64_t)mn * s_mn;
                const double x = row[min(k, K - 1)];
                const double y = row[min(k + 1, K - 1)];
                r[p] = d2{k < K ? x : 0.0, k + 1 < K ? y : 0.0};
            }
        }
    } else {
        constexpr int VPR = BMN / 2; // vectors per k-row
        const bool full_mn = (mn0 + BMN <= MN);
        if (full_k && !full_mn && MN >= 2) {
            // edge tile along mn, interior along k: clamp the vector to the last full pair and shift
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int v = min(tid + p * NT, NVEC - 1);
                const int k = k0 + v / VPR;
                const int mn = mn0 + 2 * (v % VPR);
                const d2 t = *(gcptr2)(base + (int64_t)k * s_k + min(mn, MN - 2));
                r[p] = (mn == MN - 1) ? d2{t.y, 0.0} : t;
            }
        } else if (full_k && full_mn) {
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int v = min(tid + p * NT, NVEC - 1);
                const int k = k0 + v / VPR;
                const int mn = mn0 + 2 * (v % VPR);
                r[p] = *(gcptr2)(base + (int64_t)k * s_k + mn);
            }
        } else {
#pragma unroll
            for (int p = 0; p < NV; ++p) {
                const int v = min(tid + p * NT, NVEC - 1);
                const int k = k0 + v / VPR;
                const int mn = mn0 + 2 * (v % VPR);
                gcptr row = base + (int64_t)min(k, K - 1) * s_k;
                const double x = row[min(mn, MN - 1)];
                const double y = row[min(mn + 1, MN - 1)];
                r[p] = d2{k < K ? x : 0.0, k < K ? y : 0.0};
            }
        }
    }
}

template <int BMN, int NT>
__device__ __forceinline__ void store_tile(const d2 (&r)[(BMN * 8 + NT - 1) / NT], double* __restrict__ lds,
                                           bool k_contig, int tid)
{
    constexpr int NV = (BMN * 8 + NT - 1) / NT;
    constexpr int NVEC = BMN * 8; // vectors in the tile (threads beyond them idle: NT > NVEC for the 16-wide class)
    if (k_contig) {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int v = tid + p * NT;
            if (v < NVEC) *reinterpret_cast<d2*>(lds + (v >> 3) * (BK + 2) + 2 * (v & 7)) = r[p];
        }
    } else {
        constexpr int VPR = BMN / 2;
        constexpr int LS = lds_km_stride(BMN);
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int v = tid + p * NT;
            if (v < NVEC) *reinterpret_cast<d2*>(lds + (v / VPR) * LS + 2 * (v % VPR)) = r[p];
        }
    }
}

__host__ __device__ constexpr int smem_doubles(int BM, int BN, int WGM, int WGN, int KS)
{
    const int stage = 2 * (lds_tile_doubles(BM) + lds_tile_doubles(BN));
    const int red = (KS > 1) ? KS * WGM * WGN * (BM / WGM / 16) * (BN / WGN / 16) * 256 : 0;
    return stage > red ? stage : red;
}
constexpr int kSmemDoubles = smem_doubles(128, 128, 2, 2, 1); // the largest class

// KS > 1: the waves are additionally split along K (wave group g takes the k-steps kk with
// kk % KS == g of every staged k-tile) and the accumulators are summed through LDS at the end --
// for the narrow (<= 32 wide) tile classes, where one wave per tile would leave the SIMDs idle on
// the long-K products of the blocked QR (V^T A, K ~ 1000) and of tall-skinny blocks.
template <int BM, int BN, int WGM, int WGN, int KS = 1>
__device__ __forceinline__ void gemm_tile(const DevProb* __restrict__ probs, const DevSeg* __restrict__ segs,
                                          const DevTile t, double* __restrict__ smem)
{
    constexpr int NT = 64 * WGM * WGN * KS;
    static_assert(NT == 256, "all tile classes run in 256-thread workgroups (one launch, one queue)");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int LA = lds_tile_doubles(BM), LB = lds_tile_doubles(BN);
    static_assert(smem_doubles(BM, BN, WGM, WGN, KS) <= kSmemDoubles, "LDS budget");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kgrp = wave / (WGM * WGN);          // K-split group of this wave
    const int wtile = wave % (WGM * WGN);
    const int wm = wtile / WGN, wn = wtile % WGN;

    const DevProb pr = probs[t.prob];
    const DevSeg* gsegs = segs;
    const int row0 = t.tm, col0 = t.tn;

    d4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    d2 ra[(BM * 8 + NT - 1) / NT], rb[(BN * 8 + NT - 1) / NT];

    // cursor over (segment, k0)
    int seg = pr.seg_begin;
    // skip empty segments
    while (seg < pr.seg_end && gsegs[seg].K <= 0) ++seg;
    if (seg < pr.seg_end) {
        DevSeg sg = gsegs[seg];
        int k0 = 0;
        bool a_kc = (sg.a_cs == 1), b_kc = (sg.b_rs == 1);
        load_tile<BM, NT>(ra, sg.A, sg.a_rs, sg.a_cs, a_kc, row0, k0, pr.M, sg.K, tid);
        load_tile<BN, NT>(rb, sg.B, sg.b_cs, sg.b_rs, b_kc, col0, k0, pr.N, sg.K, tid);
        store_tile<BM, NT>(ra, smem, a_kc, tid);
        store_tile<BN, NT>(rb, smem + LA, b_kc, tid);
        __syncthreads();
        int buf = 0;
        // MFMA burst on the k-tile staged in LDS buffer `b` (layouts ca_kc / cb_kc)
        auto compute = [&](const int b, const bool ca_kc, const bool cb_kc) {
            const double* As = smem + b * (LA + LB);
            const double* Bs = As + LA;
            const int sAm = ca_kc ? (BK + 2) : 1;
            const int sAk = ca_kc ? 1 : lds_km_stride(BM);
            const int sBn = cb_kc ? (BK + 2) : 1;
            const int sBk = cb_kc ? 1 : lds_km_stride(BN);
            const double* ap = As + (wm * WM + (lane & 15)) * sAm + (lane >> 4) * sAk;
            const double* bp = Bs + (wn * WN + (lane & 15)) * sBn + (lane >> 4) * sBk;
            if (kPrioMfma) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                if (KS > 1 && (kk % KS) != kgrp) continue;
                double a[TM], b_[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = ap[i * 16 * sAm + kk * 4 * sAk];
#pragma unroll
                for (int j = 0; j < TN; ++j) b_[j] = bp[j * 16 * sBn + kk * 4 * sBk];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b_[j], acc[i][j], 0, 0, 0);
            }
            if (kPrioMfma) __builtin_amdgcn_s_setprio(0);
        };
        const bool interior = (row0 + BM <= pr.M) && (col0 + BN <= pr.N);
        const bool fast_ok = interior || (pr.M >= 2 && pr.N >= 2);
        while (true) {
            // ---- steady state: interior tile, the next k-tile is a full one of the same segment.
            // One basic block per k-tile (per-thread pointers advance by a constant, no clamps, no
            // cursor logic), so the address arithmetic does not sit between two MFMA bursts.
            if constexpr (KS == 1 && kFastLoop && (BM * 8) % NT == 0 && (BN * 8) % NT == 0) if (fast_ok && k0 + 2 * BK <= sg.K) {
                constexpr int NVA = (BM * 8) / NT, NVB = (BN * 8) / NT;
                const int n_fast = (sg.K - k0) / BK - 1;
                gcptr pa[NVA], pb[NVB];
                int oa[NVA], ob[NVB];
                bool sha[NVA], shb[NVB]; // odd edge of a vector that runs along mn: keep .y, zero the rest
                const int64_t inc_a = a_kc ? (int64_t)BK : (int64_t)BK * sg.a_cs;
                const int64_t inc_b = b_kc ? (int64_t)BK : (int64_t)BK * sg.b_rs;
                // out-of-range mn positions are clamped exactly as in load_tile (their products only
                // reach accumulator rows/columns that are never stored)
#pragma unroll
                for (int p = 0; p < NVA; ++p) {
                    const int v = tid + p * NT;
                    if (a_kc) {
                        const int mn = min(row0 + (v >> 3), pr.M - 1);
                        pa[p] = (gcptr)sg.A + (int64_t)mn * sg.a_rs + (k0 + BK + 2 * (v & 7));
                        oa[p] = (v >> 3) * (BK + 2) + 2 * (v & 7);
                        sha[p] = false;
                    } else {
                        const int mn = row0 + 2 * (v % (BM / 2));
                        pa[p] = (gcptr)sg.A + (int64_t)(k0 + BK + v / (BM / 2)) * sg.a_cs + min(mn, pr.M - 2);
                        oa[p] = (v / (BM / 2)) * lds_km_stride(BM) + 2 * (v % (BM / 2));
                        sha[p] = (mn == pr.M - 1);
                    }
                }
#pragma unroll
                for (int p = 0; p < NVB; ++p) {
                    const int v = tid + p * NT;
                    if (b_kc) {
                        const int mn = min(col0 + (v >> 3), pr.N - 1);
                        pb[p] = (gcptr)sg.B + (int64_t)mn * sg.b_cs + (k0 + BK + 2 * (v & 7));
                        ob[p] = (v >> 3) * (BK + 2) + 2 * (v & 7);
                        shb[p] = false;
                    } else {
                        const int mn = col0 + 2 * (v % (BN / 2));
                        pb[p] = (gcptr)sg.B + (int64_t)(k0 + BK + v / (BN / 2)) * sg.b_rs + min(mn, pr.N - 2);
                        ob[p] = (v / (BN / 2)) * lds_km_stride(BN) + 2 * (v % (BN / 2));
                        shb[p] = (mn == pr.N - 1);
                    }
                }
                for (int it = 0; it < n_fast; ++it) {
#pragma unroll
                    for (int p = 0; p < NVA; ++p) {
                        ra[p] = *(gcptr2)pa[p];
                        pa[p] += inc_a;
                    }
#pragma unroll
                    for (int p = 0; p < NVB; ++p) {
                        rb[p] = *(gcptr2)pb[p];
                        pb[p] += inc_b;
                    }
                    compute(buf, a_kc, b_kc);
                    buf ^= 1;
                    double* sa = smem + buf * (LA + LB);
                    if (!interior) {
#pragma unroll
                        for (int p = 0; p < NVA; ++p)
                            if (sha[p]) ra[p] = d2{ra[p].y, 0.0};
#pragma unroll
                        for (int p = 0; p < NVB; ++p)
                            if (shb[p]) rb[p] = d2{rb[p].y, 0.0};
                    }
#pragma unroll
                    for (int p = 0; p < NVA; ++p) *reinterpret_cast<d2*>(sa + oa[p]) = ra[p];
#pragma unroll
                    for (int p = 0; p < NVB; ++p) *reinterpret_cast<d2*>(sa + LA + ob[p]) = rb[p];
                    __syncthreads();
                }
                k0 += n_fast * BK;
            }
            // layout of the tile being computed
            const bool ca_kc = a_kc, cb_kc = b_kc;
            // advance the cursor to the next k-tile
            int nseg = seg, nk0 = k0 + BK;
            bool have_next = true;
            if (nk0 >= sg.K) {
                nk0 = 0;
                ++nseg;
                while (nseg < pr.seg_end && gsegs[nseg].K <= 0) ++nseg;
                have_next = nseg < pr.seg_end;
            }
            if (have_next) {
                if (nseg != seg) {
                    sg = gsegs[nseg];
                    a_kc = (sg.a_cs == 1);
                    b_kc = (sg.b_rs == 1);
                }
                load_tile<BM, NT>(ra, sg.A, sg.a_rs, sg.a_cs, a_kc, row0, nk0, pr.M, sg.K, tid);
                load_tile<BN, NT>(rb, sg.B, sg.b_cs, sg.b_rs, b_kc, col0, nk0, pr.N, sg.K, tid);
            }
            compute(buf, ca_kc, cb_kc);
            if (!have_next) break;
            buf ^= 1;
            store_tile<BM, NT>(ra, smem + buf * (LA + LB), a_kc, tid);
            store_tile<BN, NT>(rb, smem + buf * (LA + LB) + LA, b_kc, tid);
            __syncthreads();
            seg = nseg;
            k0 = nk0;
        }
    }

    if (KS > 1) { // sum the K-split partial accumulators; group 0 keeps the total
        __syncthreads();
        double* mine = smem + (size_t)(wave * TM * TN) * 256;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[((i * TN + j) * 4 + r) * 64 + lane] = acc[i][j][r];
        __syncthreads();
        if (kgrp == 0) {
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const double* other = smem + (size_t)((g * WGM * WGN + wtile) * TM * TN) * 256;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += other[((i * TN + j) * 4 + r) * 64 + lane];
        }
        }
    }
    // ---- optional left factor (classes whose tile holds all M <= 32 rows in one wave: 32x32 and 32x128).
    //      The f64 C/D register map IS the B-operand map of consecutive k-steps: row (lane>>4) + 4*reg of
    //      accumulator tile t is the operand of k-step 4*t + reg, so L * acc needs no data movement at all.
    if constexpr (BM == 32 && WGM == 1) {
        if (pr.L != nullptr && (KS == 1 || kgrp == 0)) {
            gcptr Lp = (gcptr)pr.L;
            d4 out[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) out[i][j] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int kcol = 4 * kk + (lane >> 4);
                double a[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int lrow = 16 * i + (lane & 15);
                    const bool ok = lrow < pr.M && kcol < pr.M;
                    a[i] = ok ? Lp[(int64_t)lrow * pr.l_rs + (int64_t)kcol * pr.l_cs] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        out[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], acc[kk >> 2][j][kk & 3], out[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = out[i][j];
        }
    }
    // ---- epilogue: C = alpha*acc + beta*C.  f64 MFMA C/D map: col = lane&15, row = (lane>>4)+4*reg
    if (KS == 1 || kgrp == 0) {
    const bool use_beta = (pr.beta != 0.0);
    if (use_beta) {
        // accumulating products (the rank-32 updates of the blocked QR): ALL reads of an accumulator row-block are
        // issued before its first store -- interleaved `load C; store C` pairs cannot be reordered by the compiler
        // (same array) and cost one memory latency per 16-column group: 16 round trips per tile instead of TM
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            double cv[4][TN];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wm * WM + i * 16 + (lane >> 4) + 4 * r;
                gptr crow = (gptr)(pr.C + (int64_t)min(row, pr.M - 1) * pr.ldc);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = col0 + wn * WN + j * 16 + (lane & 15);
                    cv[r][j] = crow[min(col, pr.N - 1)];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wm * WM + i * 16 + (lane >> 4) + 4 * r;
                if (row >= pr.M) continue;
                gptr crow = (gptr)(pr.C + (int64_t)row * pr.ldc);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = col0 + wn * WN + j * 16 + (lane & 15);
                    if (col < pr.N) crow[col] = pr.alpha * acc[i][j][r] + pr.beta * cv[r][j];
                }
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + wm * WM + i * 16 + (lane >> 4) + 4 * r;
            if (row >= pr.M) continue;
            gptr crow = (gptr)(pr.C + (int64_t)row * pr.ldc);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + wn * WN + j * 16 + (lane & 15);
                if (col < pr.N) crow[col] = pr.alpha * acc[i][j][r];
            }
        }
    }
    }
    } // writer waves
}

// Streaming kernel for the products of a small operator with a long operand (M <= 16, every K segment <= 32, B and C
// contiguous along n): C(M x N) = A(M x K) B(K x N) of an MPO tensor (D d x D d) with a million columns is HBM-bound
// (K + M doubles moved per column for 2 M K flops), so no LDS staging of B and no MFMA, and -- unlike the MFMA kernel,
// whose 256 VGPRs allow two waves per SIMD -- a small register budget: four waves per SIMD with eight 16-byte loads
// in flight per lane keep ~130 KB per CU outstanding.  A sits in LDS (broadcast reads), every thread owns one column
// pair, the sixteen row accumulators stay in registers.  One workgroup = one tile of 16 rows x SK_W columns.
constexpr int SK_W = 512;   // columns per tile
constexpr int SK_KMAX = 32;
__global__ void __launch_bounds__(256, 4) gemm_skinny_kernel(const DevProb* __restrict__ probs, const DevSeg* __restrict__ segs,
                                                             const DevTile* __restrict__ tiles)
{
    __shared__ double As[SK_KMAX * 16];
    const DevTile t = tiles[blockIdx.x];
    const DevProb pr = probs[t.prob];
    const int tid = threadIdx.x;
    const int n0 = t.tn + 2 * tid;
    // branch-free edges (N >= 2 here): a pair beyond the last column reads the last full pair and is not stored, the
    // odd last column reads that pair and keeps its second entry
    const int c0 = min(n0, pr.N - 2);
    const bool sh0 = (n0 == pr.N - 1);
    d2 acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = d2{0.0, 0.0};
    for (int sgi = pr.seg_begin; sgi < pr.seg_end; ++sgi) {
        const DevSeg sg = segs[sgi];
        if (sg.K <= 0) continue;
        __syncthreads(); // the previous segment's A is no longer read
        for (int e = tid; e < SK_KMAX * 16; e += 256) {
            const int k = e >> 4, m = e & 15;
            As[e] = (k < sg.K && m < pr.M) ? ((gcptr)sg.A)[(int64_t)m * sg.a_rs + (int64_t)k * sg.a_cs] : 0.0;
        }
        gcptr b0 = (gcptr)sg.B + c0;
        constexpr int PF = 8; // k-rows in flight per thread
        d2 q[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j)
            if (j < sg.K) q[j] = *(gcptr2)(b0 + (int64_t)j * sg.b_rs);
        __syncthreads();
        for (int k = 0; k < sg.K; k += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if (k + j < sg.K) {
                    d2 v = q[j];
                    if (k + j + PF < sg.K) q[j] = *(gcptr2)(b0 + (int64_t)(k + j + PF) * sg.b_rs);
                    if (sh0) v = d2{v.y, 0.0};
                    const double* ak = As + (k + j) * 16;
#pragma unroll
                    for (int m = 0; m < 16; ++m) acc[m] += ak[m] * v;
                }
            }
        }
    }
    const bool use_beta = pr.beta != 0.0;
    const bool in0 = n0 < pr.N, full0 = n0 + 1 < pr.N;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        if (m < pr.M) { // (no `break`: the unrolled loop keeps acc[] in registers)
            gptr cp = (gptr)(pr.C + (int64_t)m * pr.ldc) + n0;
            d2 r = pr.alpha * acc[m];
            if (full0) {
                if (use_beta) r += pr.beta * *(gcptr2)cp;
                *(GLOBAL_AS d2u*)cp = r;
            } else if (in0) {
                cp[0] = use_beta ? r.x + pr.beta * cp[0] : r.x;
            }
        }
    }
}

// out-of-line instances for the small classes: they keep their own (small) register budget instead
// of inflating the 128x128 path, which sits right at the 256-VGPR / 2-waves-per-SIMD limit
template <int BM, int BN, int WGM, int WGN, int KS>
__device__ __noinline__ void gemm_tile_ool(const DevProb* __restrict__ probs, const DevSeg* __restrict__ segs,
                                           const DevTile t, double* __restrict__ smem)
{
    gemm_tile<BM, BN, WGM, WGN, KS>(probs, segs, t, smem);
}

// ONE launch for the whole block list: persistent 256-thread workgroups pull tiles of all four
// classes from a single queue sorted by work (dynamic LPT), so the few small-class tiles fill the
// gaps instead of running as under-filled launches of their own (chi=4096: 53 us of 405).
__global__ void __launch_bounds__(256, 2)
gemm_grouped_kernel(const DevProb* __restrict__ probs, const DevSeg* __restrict__ segs,
                    const DevTile* __restrict__ tiles, const int n_tiles, unsigned int* __restrict__ counter)
{
    __shared__ __attribute__((aligned(16))) double smem[kSmemDoubles];
    __shared__ int s_tile;
    // The first tile of a workgroup is its own index (the queue is sorted by work, so this IS the head of the queue);
    // only later tiles are drawn from the counter.  A launch with no more tiles than workgroups -- every panel step of
    // the blocked QR, the small lists -- then runs without a single atomic: 512 workgroups drawing from one address at
    // the start and once more to find the queue empty cost several microseconds of a 40 us launch.
    int tile_idx = (int)blockIdx.x;
    const bool one_round = n_tiles <= (int)gridDim.x;
    for (;;) {
        if (tile_idx >= n_tiles) break;
        const DevTile t = tiles[tile_idx];
        switch (t.pad) { // tile class
        case 0: gemm_tile<128, 128, 2, 2>(probs, segs, t, smem); break;
        case 1: gemm_tile_ool<64, 64, 2, 2, 1>(probs, segs, t, smem); break;
        case 2: gemm_tile_ool<32, 32, 1, 1, 4>(probs, segs, t, smem); break;
        case 4: gemm_tile_ool<128, 64, 2, 2, 1>(probs, segs, t, smem); break;
        case 5: gemm_tile_ool<16, 128, 1, 4, 1>(probs, segs, t, smem); break;
        case 6: gemm_tile_ool<128, 16, 4, 1, 1>(probs, segs, t, smem); break;
        case 7: gemm_tile_ool<32, 128, 1, 4, 1>(probs, segs, t, smem); break;
        case 8: gemm_tile_ool<128, 32, 4, 1, 1>(probs, segs, t, smem); break;
        case 9: gemm_tile_ool<64, 128, 2, 2, 1>(probs, segs, t, smem); break;
        default: gemm_tile_ool<16, 16, 1, 1, 4>(probs, segs, t, smem); break;
        }
        if (one_round) break;
        if (threadIdx.x == 0) s_tile = (int)gridDim.x + (int)atomicAdd(counter, 1u);
        __syncthreads();
        tile_idx = s_tile;
        __syncthreads(); // s_tile is rewritten next round; also fences the LDS tiles of the previous tile
    }
}

// ---- MFMA f64 issue-rate micro-benchmark -------------------------------------------------------
template <int NACC>
__global__ void __launch_bounds__(256) mfma_f64_peak_kernel(double* out, int iters)
{
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    // PEAK_UNROLL rounds per trip: the compiler keeps the accumulators of the loop-carried d4's in VGPRs and copies them to and
    // from the AGPRs the MFMA uses at every trip (32 v_accvgpr_write + 32 v_accvgpr_read + s_nop 13 around FOUR MFMAs in the
    // round-1 form of this loop, which is why it read 47 TFLOP/s: PMC showed 2.39 GHz and MFMA busy 0.61, i.e. an issue-limited
    // loop, not a throttled clock -- scripts/mfma_peak_probe.py); unrolled, the copies amortise over PEAK_UNROLL * NACC MFMAs
    constexpr int PEAK_UNROLL = 16;
    for (int it = 0; it < iters; it += PEAK_UNROLL) {
#pragma unroll
        for (int u = 0; u < PEAK_UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) // (accumulators pinned to AGPRs: no copies inside the trip)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    d4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

struct TileClass {
    int bm, bn; // tile shape
};
// class 4 (128 x 64) is class 0 with the N direction cut in half: same per-wave K loop depth, twice
// as many tiles for the queue to balance
// classes 5..8: strips for skinny problems (one extent below 40, the other long): an m x 5 x 5
// product of an MPO tensor with a million columns would otherwise shatter into 16 x 16 tiles
// class 10: the tiles of gemm_skinny_kernel (16 rows x SK_W columns, no MFMA; a launch of its own, tile list 1)
constexpr int kNumClasses = 11;
constexpr TileClass kClasses[kNumClasses] = {{128, 128}, {64, 64}, {32, 32}, {16, 16}, {128, 64}, {16, 128},
                                             {128, 16}, {32, 128}, {128, 32}, {64, 128}, {16, SK_W}};

inline int pick_class(int64_t M, int64_t N)
{
    const int64_t s = std::min(M, N);
    const int64_t l = std::max(M, N);
    if (s >= 96 && l >= 128) return 0;
    if (s >= 40) return 1;
    if (l >= 256) { // skinny: a strip along the long extent
        if (s >= 20) return M <= N ? 7 : 8;
        return M <= N ? 5 : 6;
    }
    if (s >= 20) return 2;
    return 3;
}

} // namespace

struct cyb_gemm_plan_s {
    int device = 0;
    void* dev_blob = nullptr; // probs | segs | tiles of all classes
    DevProb* d_probs = nullptr;
    DevSeg* d_segs = nullptr;
    DevTile* d_tiles[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned int* d_counters = nullptr; // one tile-queue head per class
    int64_t n_tiles[4] = {0, 0, 0, 0};
    double flops = 0, bytes = 0;
};

namespace {

struct HostBlob {
    std::vector<char> data;
    size_t off_p = 0, off_s = 0, off_t[4] = {0, 0, 0, 0}, off_c = 0;
    int64_t n_tiles[4] = {0, 0, 0, 0};
    double flops = 0, bytes = 0;
};

// Validate the problem list and build the device image (descriptors + tile queues).
int build_blob(const cyb_gemm_prob* probs, int64_t n_probs, const cyb_gemm_seg* segs, int64_t n_segs, HostBlob& hb,
               int n_cu_hint = 256, const cyb::GemmPost* post = nullptr, bool allow_skinny = true)
{
    CYB_REQUIRE(n_probs >= 0 && n_segs >= 0, "gemm: negative count");
    CYB_REQUIRE(n_probs == 0 || probs, "gemm: probs is NULL");
    CYB_REQUIRE(n_segs == 0 || segs, "gemm: segs is NULL");
    CYB_REQUIRE(n_probs < (1ll << 31) && n_segs < (1ll << 31), "gemm: too many problems");

    std::vector<DevProb> hp((size_t)n_probs);
    std::vector<DevSeg> hs((size_t)n_segs);
    double flops = 0, bytes = 0;
    for (int64_t s = 0; s < n_segs; ++s) {
        const cyb_gemm_seg& g = segs[s];
        CYB_REQUIRE(g.K >= 0 && g.K < (1ll << 31), "gemm segment %lld: bad K=%lld", (long long)s, (long long)g.K);
        CYB_REQUIRE(g.K == 0 || (g.A && g.B), "gemm segment %lld: NULL operand", (long long)s);
        hs[(size_t)s] = DevSeg{g.A, g.B, g.a_rs, g.a_cs, g.b_rs, g.b_cs, (int32_t)g.K, 0};
    }
    struct HostTile {
        DevTile t;
        int64_t work;
    };
    std::vector<HostTile> ht[kNumClasses];
    // Granularity: with fewer 128x128 tiles than CUs the chip is not even filled once; 64x64 tiles
    // give the dynamic queue four times as many pieces (chi=1024 theta: 81 -> 44 us).  Above that
    // the 128x128 class wins on per-tile efficiency (55 vs 34 TFLOP/s on uniform 4096^3).
    int64_t n128 = 0;
    for (int64_t p = 0; p < n_probs; ++p)
        if (probs[p].M > 0 && probs[p].N > 0 && pick_class(probs[p].M, probs[p].N) == 0)
            n128 += cdiv64(probs[p].M, 128) * cdiv64(probs[p].N, 128);
    const bool demote = n128 > 0 && n128 < (int64_t)n_cu_hint; // fewer than one 128-tile per CU
    static const int split_env = getenv("CYB_GEMM_SPLITN") ? atoi(getenv("CYB_GEMM_SPLITN")) : 0;  // measured: 128x64 tiles lose more per-tile efficiency than they gain in balance
    const bool split_n = !demote && n128 < (int64_t)split_env * 2 * n_cu_hint; // few tiles per slot: halve them
    static const bool ragged_env = !(getenv("CYB_GEMM_RAGGED") && atoi(getenv("CYB_GEMM_RAGGED")) == 0);
    for (int64_t p = 0; p < n_probs; ++p) {
        const cyb_gemm_prob& q = probs[p];
        CYB_REQUIRE(q.M >= 0 && q.N >= 0 && q.M < (1ll << 31) && q.N < (1ll << 31),
                    "gemm problem %lld: bad shape %lld x %lld", (long long)p, (long long)q.M, (long long)q.N);
        CYB_REQUIRE(q.seg_begin >= 0 && q.seg_begin <= q.seg_end && q.seg_end <= n_segs,
                    "gemm problem %lld: bad segment range [%d,%d)", (long long)p, q.seg_begin, q.seg_end);
        CYB_REQUIRE((q.M == 0 || q.N == 0) || q.C, "gemm problem %lld: C is NULL", (long long)p);
        CYB_REQUIRE(q.ldc >= q.N, "gemm problem %lld: ldc=%lld < N=%lld", (long long)p, (long long)q.ldc, (long long)q.N);
        int64_t ktot = 0;
        for (int32_t s = q.seg_begin; s < q.seg_end; ++s) {
            const cyb_gemm_seg& g = segs[s];
            if (g.K == 0) continue;
            // every operand view must be contiguous along one of its two directions (a 1 x K or
            // M x 1 view is contiguous whatever the stride of its singleton direction is)
            int64_t a_rs = g.a_rs, a_cs = g.a_cs, b_rs = g.b_rs, b_cs = g.b_cs;
            if (a_cs != 1 && a_rs != 1) {
                if (g.K == 1) a_cs = 1;
                else if (q.M == 1) a_rs = 1;
            }
            if (b_cs != 1 && b_rs != 1) {
                if (q.N == 1) b_cs = 1;
                else if (g.K == 1) b_rs = 1;
            }
            CYB_REQUIRE(a_cs == 1 || a_rs == 1, "gemm segment %d: A view has no unit stride (%lld,%lld)", s,
                        (long long)g.a_rs, (long long)g.a_cs);
            CYB_REQUIRE(b_cs == 1 || b_rs == 1, "gemm segment %d: B view has no unit stride (%lld,%lld)", s,
                        (long long)g.b_rs, (long long)g.b_cs);
            hs[(size_t)s].a_rs = a_rs;
            hs[(size_t)s].a_cs = a_cs;
            hs[(size_t)s].b_rs = b_rs;
            hs[(size_t)s].b_cs = b_cs;
            ktot += g.K;
            flops += 2.0 * (double)q.M * (double)q.N * (double)g.K;
            bytes += 8.0 * ((double)q.M * g.K + (double)g.K * q.N);
        }
        bytes += 8.0 * (double)q.M * (double)q.N * (q.beta != 0.0 ? 2.0 : 1.0);
        hp[(size_t)p] = DevProb{q.C, q.ldc, (int32_t)q.M, (int32_t)q.N, q.seg_begin, q.seg_end, q.alpha, q.beta, nullptr, 0, 0};
        const bool has_post = post && post[p].L;
        if (has_post) {
            CYB_REQUIRE(q.M <= 32, "gemm problem %lld: a left factor needs M <= 32 (M=%lld)", (long long)p, (long long)q.M);
            hp[(size_t)p].L = post[p].L;
            hp[(size_t)p].l_rs = (int32_t)post[p].rs;
            hp[(size_t)p].l_cs = (int32_t)post[p].cs;
        }
        if (q.M == 0 || q.N == 0) continue;
        int c = pick_class(q.M, q.N);
        static const int post_cut = getenv("CYB_GEMM_POSTCUT") ? atoi(getenv("CYB_GEMM_POSTCUT")) : 256;
        if (has_post) c = q.N >= post_cut ? 7 : 2; // a class whose tile holds all rows in one wave
        static const bool skinny_env = !(getenv("CYB_GEMM_SKINNY") && atoi(getenv("CYB_GEMM_SKINNY")) == 0);
        if (allow_skinny && skinny_env && !has_post && q.M <= 16 && q.N >= 4 * SK_W) {
            bool ok = true;
            for (int32_t sg = q.seg_begin; sg < q.seg_end && ok; ++sg)
                ok = segs[sg].K == 0 || (segs[sg].K <= SK_KMAX && hs[(size_t)sg].b_cs == 1);
            if (ok) c = 10;
        }
        if (c == 0 && demote) c = 1;
        if (c == 0 && split_n) c = 4;
        if (c <= 1 && ragged_env) {
            // Ragged edges: full tiles of the base size, then ONE narrower segment per direction that
            // matches the remainder (64, or a 32 / 16 strip), instead of padding the last tile row and
            // column to the base size (U(1)xU(1) chi=4096 list: useful / executed flops 0.58 -> 0.86).
            const int base = kClasses[c].bm;
            auto cut = [base](int64_t ext, std::vector<std::pair<int32_t, int>>& out) {
                out.clear();
                int64_t off = 0;
                for (; off + base <= ext; off += base) out.push_back({(int32_t)off, base});
                const int64_t r = ext - off;
                if (r == 0) return;
                int w = base;
                if (off > 0 || base == 64) { // a remainder after full tiles (or the demoted 64 base)
                    if (r <= 16) w = 16;
                    else if (r <= 32) w = 32;
                    else if (r <= 64) w = 64;
                }
                out.push_back({(int32_t)off, std::min(w, base)});
            };
            std::vector<std::pair<int32_t, int>> rs, cs;
            cut(q.M, rs);
            cut(q.N, cs);
            for (auto& r : rs)
                for (auto& cc : cs) {
                    int bm = r.second, bn = cc.second, cls = -1;
                    for (int k = 0; k < kNumClasses; ++k)
                        if (kClasses[k].bm == bm && kClasses[k].bn == bn) cls = k;
                    if (cls < 0) { // no such rectangle: the covering square (only corner tiles get here)
                        const int mx = std::max(bm, bn);
                        for (int k = 0; k < 4; ++k)
                            if (kClasses[k].bm == mx) cls = k;
                    }
                    ht[cls].push_back(HostTile{DevTile{(int32_t)p, r.first, cc.first, 0}, ktot});
                }
            continue;
        }
        const int bm = kClasses[c].bm, bn = kClasses[c].bn;
        const int64_t ntm = cdiv64(q.M, bm), ntn = cdiv64(q.N, bn);
        for (int64_t tm = 0; tm < ntm; ++tm)
            for (int64_t tn = 0; tn < ntn; ++tn)
                ht[c].push_back(HostTile{DevTile{(int32_t)p, (int32_t)(tm * bm), (int32_t)(tn * bn), 0}, ktot});
    }
    // longest-K tiles first (LPT): the tail of the launch is then made of the short tiles
    // one queue for all classes, heaviest tiles first (work ~ tile area x K); it is stored as class 0
    {
        std::vector<HostTile> all, skinny;
        skinny.swap(ht[10]);
        for (auto& h : skinny) h.t.pad = 10;
        for (int c = 0; c < kNumClasses; ++c) {
            for (auto& h : ht[c]) {
                h.t.pad = c;
                h.work *= (int64_t)kClasses[c].bm * kClasses[c].bn;
                all.push_back(h);
            }
            ht[c].clear();
        }
        auto by_work = [](const HostTile& a, const HostTile& b) { return a.work > b.work; };
        std::stable_sort(all.begin(), all.end(), by_work);
        // Tail granularity: the block lists of this path give only one to two 128x128 tiles per
        // workgroup slot, so the last, partly filled round of the queue decides the makespan.  The
        // tiles beyond the last full round are cut into 128x64 halves (class 4: same per-wave k loop,
        // half the work), which lets the queue level the tail (chi=4096 theta list, 648 tiles on 512
        // slots: 309 -> 267 us; a list whose last round is full is left alone).
        static const int tail_env = getenv("CYB_GEMM_TAILSPLIT") ? atoi(getenv("CYB_GEMM_TAILSPLIT")) : 1;
        const size_t slots = 2 * (size_t)n_cu_hint;
        const size_t last_round = all.size() > slots ? all.size() - (all.size() - 1) / slots * slots : 0;
        if (tail_env && last_round > 0 && last_round <= slots * 3 / 4) { // a full last round needs no levelling
            const size_t keep = all.size() - last_round; // full rounds stay as they are
            std::vector<HostTile> tail;
            for (size_t i = keep; i < all.size(); ++i) {
                const HostTile& h = all[i];
                if (h.t.pad != 0) {
                    tail.push_back(h);
                    continue;
                }
                const int64_t N = probs[h.t.prob].N;
                // a short last round (at most 3/8 of the slots) is cut into 128x32 quarters instead (class 8), so that
                // its pieces still fill the chip: 136 tiles -> 544 pieces on 512 slots rather than 272 halves
                const int parts = (tail_env >= 2 && last_round <= slots * 3 / 8) ? 4 : 2;
                for (int part = 0; part < parts; ++part) {
                    const int32_t tn = h.t.tn + (128 / parts) * part; // first column of the piece
                    if ((int64_t)tn >= N) continue;
                    tail.push_back(HostTile{DevTile{h.t.prob, h.t.tm, tn, parts == 4 ? 8 : 4}, h.work / parts});
                }
            }
            all.resize(keep);
            std::stable_sort(tail.begin(), tail.end(), by_work);
            all.insert(all.end(), tail.begin(), tail.end());
        }
        // XCD-aware placement of a problem's tiles.  Workgroups are dealt round-robin over the 8 XCDs (queue position
        // i -> workgroup i -> XCD i % 8 for the first round; speed only, never correctness) and every XCD has its own
        // L2.  The work-sorted queue keeps the tiles of one problem in one run, so without care the tiles that share an
        // A row panel or a B column panel land on eight different L2s and every panel is fetched once per tile (counter
        // traffic 4.2x the algorithmic bytes on the chi=4096 list, L2 hit rate 0.39: profiles/r01_gemm_pmc_summary.json).
        // Inside each run the tiles are therefore re-dealt: a snake over bands of two tile rows orders them so that
        // neighbours share a panel, the snake is cut into eight compact groups, and group x goes to the positions with
        // i % 8 == x.  A group of ~6 tiles of a 7 x 6 problem then fetches ~5 panels instead of 12, while the problem
        // still spreads over all eight L2s (putting a whole problem on ONE XCD was measured slower in round 1: all of its
        // tiles then request the same lines at the same moment).
        static const int xcd_env = getenv("CYB_GEMM_XCD") ? atoi(getenv("CYB_GEMM_XCD")) : 2;
        if (xcd_env && all.size() >= 16) {
            constexpr int NX = 8;
            struct Run {
                size_t s, e;
            };
            std::vector<Run> runs;
            for (size_t s0 = 0; s0 < all.size();) {
                size_t e0 = s0 + 1;
                while (e0 < all.size() && all[e0].t.prob == all[s0].t.prob && all[e0].t.pad == all[s0].t.pad) ++e0;
                runs.push_back(Run{s0, e0});
                s0 = e0;
            }
            auto snake = [&](std::vector<HostTile>& run) { // a snake over bands of two tile rows: neighbours share a panel
                const int bm = kClasses[run[0].t.pad].bm, bn = kClasses[run[0].t.pad].bn;
                auto key = [bm, bn](const HostTile& h) {
                    const int r = h.t.tm / bm, c = h.t.tn / bn;
                    const int band = r / 2;
                    const int cc = (band & 1) ? (1 << 20) - c : c; // odd bands run backwards
                    return ((int64_t)band << 42) | ((int64_t)cc << 21) | (int64_t)((c & 1) ? 1 - (r & 1) : (r & 1));
                };
                std::stable_sort(run.begin(), run.end(), [&](const HostTile& a, const HostTile& b) { return key(a) < key(b); });
            };
            // deal the tiles of `run` (snake order) to the positions of [s, e) whose XCD class lies in [x0, x0 + nx)
            auto deal = [&](std::vector<HostTile>& run, size_t s, size_t e, int x0, int nx) {
                snake(run);
                std::vector<size_t> cnt((size_t)NX, 0), beg((size_t)NX, 0), used((size_t)NX, 0);
                for (size_t i = s; i < e; ++i) {
                    const int x = (int)(i % NX);
                    if (x >= x0 && x < x0 + nx) ++cnt[(size_t)x];
                }
                size_t acc = 0;
                for (int k = 0; k < nx; ++k) {
                    const int x = x0 + k;
                    beg[(size_t)x] = acc;
                    acc += cnt[(size_t)x];
                }
                for (size_t i = s; i < e; ++i) {
                    const int x = (int)(i % NX);
                    if (x >= x0 && x < x0 + nx) all[i] = run[beg[(size_t)x] + used[(size_t)x]++];
                }
            };
            std::vector<HostTile> ra, rb;
            for (size_t r = 0; r < runs.size(); ++r) {
                const size_t L = runs[r].e - runs[r].s;
                if (L < 4) continue;
                // Two consecutive runs of the same length, class and work (the +q / -q sectors of a symmetric leg give equal
                // problems) share their range: each is dealt over FOUR XCDs, so a panel is fetched by four L2s at most
                // instead of eight, while the range as a whole still covers all eight.
                const bool pair = xcd_env >= 2 && r + 1 < runs.size() && runs[r + 1].e - runs[r + 1].s == L && L >= 8 && L % 2 == 0 &&
                                  all[runs[r].s].work == all[runs[r + 1].s].work && all[runs[r].s].t.pad == all[runs[r + 1].s].t.pad;
                if (pair) {
                    ra.assign(all.begin() + (long)runs[r].s, all.begin() + (long)runs[r].e);
                    rb.assign(all.begin() + (long)runs[r + 1].s, all.begin() + (long)runs[r + 1].e);
                    // positions of the merged range with class < 4 number exactly L when the range starts at a multiple of 8 ...
                    size_t lo = 0;
                    for (size_t i = runs[r].s; i < runs[r + 1].e; ++i) lo += (i % NX) < NX / 2 ? 1 : 0;
                    if (lo == L) { // ... otherwise the halves would not match the runs: fall back to one run over eight
                        deal(ra, runs[r].s, runs[r + 1].e, 0, NX / 2);
                        deal(rb, runs[r].s, runs[r + 1].e, NX / 2, NX / 2);
                        ++r;
                        continue;
                    }
                }
                ra.assign(all.begin() + (long)runs[r].s, all.begin() + (long)runs[r].e);
                deal(ra, runs[r].s, runs[r].e, 0, NX);
            }
        }
        ht[0].swap(all);
        ht[1].swap(skinny); // tile list 1: the streaming kernel's launch
    }
    hb.flops = flops;
    hb.bytes = bytes;
    hb.off_p = 0;
    hb.off_s = hb.off_p + sizeof(DevProb) * (size_t)n_probs;
    size_t total = hb.off_s + sizeof(DevSeg) * (size_t)n_segs;
    for (int c = 0; c < 4; ++c) {
        hb.off_t[c] = total;
        total += sizeof(DevTile) * ht[c].size();
        hb.n_tiles[c] = (int64_t)ht[c].size();
    }
    total = (total + 15) / 16 * 16;
    hb.off_c = total;
    total += 16; // four zero-initialised queue heads
    hb.data.assign(total ? total : 8, 0);
    if (n_probs) memcpy(hb.data.data() + hb.off_p, hp.data(), sizeof(DevProb) * (size_t)n_probs);
    if (n_segs) memcpy(hb.data.data() + hb.off_s, hs.data(), sizeof(DevSeg) * (size_t)n_segs);
    for (int c = 0; c < 4; ++c) {
        DevTile* dst = reinterpret_cast<DevTile*>(hb.data.data() + hb.off_t[c]);
        for (size_t i = 0; i < ht[c].size(); ++i) dst[i] = ht[c][i].t;
    }
    return CYB_OK;
}

int launch_classes(hipStream_t st, int n_cu, const DevProb* d_probs, const DevSeg* d_segs, DevTile* const d_tiles[4],
                   const int64_t n_tiles[4], unsigned int* counters)
{
    // persistent grid: two workgroups per CU (two waves per SIMD keep the MFMA pipe paced)
    const int64_t slots = 2 * (int64_t)n_cu;
    if (n_tiles[0])
        hipLaunchKernelGGL(gemm_grouped_kernel, dim3((unsigned)std::min(n_tiles[0], slots)), dim3(256), 0, st, d_probs,
                           d_segs, d_tiles[0], (int)n_tiles[0], counters);
    if (n_tiles[1])
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3((unsigned)n_tiles[1]), dim3(256), 0, st, d_probs, d_segs, d_tiles[1]);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

} // namespace

namespace cyb {
// Internal asynchronous form used by the decompositions: descriptors travel through the context's
// upload ring (no hipMalloc, no host synchronisation).
int gemm_launch_async(cyb_ctx_t ctx, const cyb_gemm_prob* probs, int64_t n_probs, const cyb_gemm_seg* segs, int64_t n_segs,
                      const GemmPost* post)
{
    if (n_probs == 0) return CYB_OK;
    HostBlob hb;
    CYB_TRY(build_blob(probs, n_probs, segs, n_segs, hb, ctx->n_cu, post));
    void* d = nullptr;
    CYB_TRY(ctx->upload(hb.data.data(), hb.data.size(), &d));
    char* base = static_cast<char*>(d);
    DevTile* tiles[4];
    for (int c = 0; c < 4; ++c) tiles[c] = reinterpret_cast<DevTile*>(base + hb.off_t[c]);
    hipEvent_t e0 = ctx->time_start, e1 = ctx->time_stop;
    ctx->time_start = ctx->time_stop = nullptr;
    if (e0) CYB_HIP(hipEventRecord(e0, ctx->stream));
    const int rc = launch_classes(ctx->stream, ctx->n_cu, reinterpret_cast<const DevProb*>(base + hb.off_p),
                                  reinterpret_cast<const DevSeg*>(base + hb.off_s), tiles, hb.n_tiles,
                                  reinterpret_cast<unsigned int*>(base + hb.off_c));
    if (e1) CYB_HIP(hipEventRecord(e1, ctx->stream));
    return rc;
}
int gemm_stage(cyb_ctx_t ctx, const cyb_gemm_prob* probs, int64_t n_probs, const cyb_gemm_seg* segs, int64_t n_segs,
               const GemmPost* post, std::vector<char>& image, GemmStaged& st)
{
    st = GemmStaged{};
    if (n_probs == 0) return CYB_OK;
    HostBlob hb;
    CYB_TRY(build_blob(probs, n_probs, segs, n_segs, hb, ctx->n_cu, post, false)); // a staged launch is ONE kernel
    const size_t off = (image.size() + 255) / 256 * 256;
    image.resize(off + hb.data.size(), 0);
    memcpy(image.data() + off, hb.data.data(), hb.data.size());
    st.offset = off;
    st.off_p = hb.off_p;
    st.off_s = hb.off_s;
    st.off_t = hb.off_t[0];
    st.off_c = hb.off_c;
    st.n_tiles = hb.n_tiles[0];
    return CYB_OK;
}

int gemm_launch_staged(cyb_ctx_t ctx, const GemmStaged& st, void* dev_image, hipStream_t stream, int n_cu)
{
    if (st.n_tiles == 0) return CYB_OK;
    char* base = static_cast<char*>(dev_image) + st.offset;
    DevTile* tiles[4] = {reinterpret_cast<DevTile*>(base + st.off_t), nullptr, nullptr, nullptr};
    const int64_t n_tiles[4] = {st.n_tiles, 0, 0, 0};
    return launch_classes(stream ? stream : ctx->stream, n_cu > 0 ? n_cu : ctx->n_cu, reinterpret_cast<const DevProb*>(base + st.off_p),
                          reinterpret_cast<const DevSeg*>(base + st.off_s), tiles, n_tiles,
                          reinterpret_cast<unsigned int*>(base + st.off_c));
}
} // namespace cyb

extern "C" {

int cyb_gemm_plan_create(cyb_ctx_t ctx, cyb_gemm_plan_t* out, const cyb_gemm_prob* probs,
                         int64_t n_probs, const cyb_gemm_seg* segs, int64_t n_segs)
{
    CYB_REQUIRE(ctx && out, "cyb_gemm_plan_create: NULL argument");
    HostBlob hb;
    CYB_TRY(build_blob(probs, n_probs, segs, n_segs, hb, ctx->n_cu));
    cyb_gemm_plan_s* pl = new cyb_gemm_plan_s();
    pl->device = ctx->device;
    pl->flops = hb.flops;
    pl->bytes = hb.bytes;
    for (int c = 0; c < 4; ++c) pl->n_tiles[c] = hb.n_tiles[c];
    hipError_t e = hipMalloc(&pl->dev_blob, hb.data.size());
    if (e != hipSuccess) {
        delete pl;
        cyb::set_error("cyb_gemm_plan_create: hipMalloc(%zu) failed: %s", hb.data.size(), hipGetErrorString(e));
        return CYB_ERR_NOMEM;
    }
    // plan creation is synchronous (it is outside any timed / captured region by contract)
    e = hipMemcpy(pl->dev_blob, hb.data.data(), hb.data.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(pl->dev_blob);
        delete pl;
        cyb::set_error("cyb_gemm_plan_create: hipMemcpy failed: %s", hipGetErrorString(e));
        return CYB_ERR_HIP;
    }
    char* d = static_cast<char*>(pl->dev_blob);
    pl->d_probs = reinterpret_cast<DevProb*>(d + hb.off_p);
    pl->d_segs = reinterpret_cast<DevSeg*>(d + hb.off_s);
    for (int c = 0; c < 4; ++c) pl->d_tiles[c] = reinterpret_cast<DevTile*>(d + hb.off_t[c]);
    pl->d_counters = reinterpret_cast<unsigned int*>(d + hb.off_c);
    *out = pl;
    return CYB_OK;
}

int cyb_gemm_plan_run(cyb_ctx_t ctx, cyb_gemm_plan_t pl)
{
    CYB_REQUIRE(ctx && pl, "cyb_gemm_plan_run: NULL argument");
    CYB_HIP(hipMemsetAsync(pl->d_counters, 0, 16, ctx->stream)); // rewind the tile queues
    return launch_classes(ctx->stream, ctx->n_cu, pl->d_probs, pl->d_segs, pl->d_tiles, pl->n_tiles, pl->d_counters);
}

int cyb_gemm_plan_destroy(cyb_gemm_plan_t pl)
{
    if (!pl) return CYB_OK;
    if (pl->dev_blob) {
        // the plan may still be in use by an enqueued kernel
        (void)hipDeviceSynchronize();
        (void)hipFree(pl->dev_blob);
    }
    delete pl;
    return CYB_OK;
}

int cyb_gemm_plan_info(cyb_gemm_plan_t pl, double* flops, double* bytes, int64_t* n_tiles, int32_t* n_launches)
{
    CYB_REQUIRE(pl, "cyb_gemm_plan_info: plan is NULL");
    if (flops) *flops = pl->flops;
    if (bytes) *bytes = pl->bytes;
    int64_t nt = 0;
    int32_t nl = 0;
    for (int c = 0; c < 4; ++c) {
        nt += pl->n_tiles[c];
        nl += pl->n_tiles[c] ? 1 : 0;
    }
    if (n_tiles) *n_tiles = nt;
    if (n_launches) *n_launches = nl;
    return CYB_OK;
}

int cyb_gemm_grouped_f64(cyb_ctx_t ctx, const cyb_gemm_prob* probs, int64_t n_probs, const cyb_gemm_seg* segs,
                         int64_t n_segs)
{
    cyb_gemm_plan_t pl = nullptr;
    CYB_TRY(cyb_gemm_plan_create(ctx, &pl, probs, n_probs, segs, n_segs));
    int st = cyb_gemm_plan_run(ctx, pl);
    cyb_gemm_plan_destroy(pl); // synchronises
    return st;
}

int cyb_gemm_grouped_enqueue_f64(cyb_ctx_t ctx, const cyb_gemm_prob* probs, int64_t n_probs, const cyb_gemm_seg* segs,
                                 int64_t n_segs)
{
    CYB_REQUIRE(ctx, "cyb_gemm_grouped_enqueue_f64: ctx is NULL");
    return cyb::gemm_launch_async(ctx, probs, n_probs, segs, n_segs, nullptr);
}

int cyb_mfma_f64_peak(cyb_ctx_t ctx, int iters, int waves_per_simd, double* tflops, double* ms_out)
{
    CYB_REQUIRE(ctx && tflops, "cyb_mfma_f64_peak: NULL argument");
    // waves_per_simd encodes (#independent accumulators)*100 + waves per SIMD; accumulators default 4
    int nacc = waves_per_simd / 100;
    waves_per_simd %= 100;
    if (nacc == 0) nacc = 4;
    CYB_REQUIRE(iters > 0 && waves_per_simd >= 1 && waves_per_simd <= 8, "cyb_mfma_f64_peak: bad arguments");
    CYB_REQUIRE(nacc == 1 || nacc == 2 || nacc == 4 || nacc == 8, "cyb_mfma_f64_peak: accumulators must be 1,2,4,8");
    const int blocks = ctx->n_cu * waves_per_simd; // 256 threads = 4 waves = one per SIMD
    double* out = nullptr;
    CYB_HIP(hipMalloc(&out, sizeof(double) * 256 * (size_t)blocks));
    hipEvent_t e0, e1;
    CYB_HIP(hipEventCreate(&e0));
    CYB_HIP(hipEventCreate(&e1));
    const int loops = std::max(16, iters / nacc / 16 * 16); // (a multiple of the kernel's unroll)
    auto launch = [&]() {
        switch (nacc) {
        case 1: hipLaunchKernelGGL(mfma_f64_peak_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, out, loops); break;
        case 2: hipLaunchKernelGGL(mfma_f64_peak_kernel<2>, dim3(blocks), dim3(256), 0, ctx->stream, out, loops); break;
        case 4: hipLaunchKernelGGL(mfma_f64_peak_kernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, out, loops); break;
        default: hipLaunchKernelGGL(mfma_f64_peak_kernel<8>, dim3(blocks), dim3(256), 0, ctx->stream, out, loops); break;
        }
    };
    launch(); // warm
    CYB_HIP(hipEventRecord(e0, ctx->stream));
    launch();
    CYB_HIP(hipEventRecord(e1, ctx->stream));
    CYB_HIP(hipEventSynchronize(e1));
    float ms = 0;
    CYB_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double n_mfma = (double)loops * nacc * 4.0 * blocks; // per wave x 4 waves x blocks
    *tflops = n_mfma * 2048.0 / (ms * 1e-3) / 1e12;
    if (ms_out) *ms_out = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    CYB_HIP(hipFree(out));
    return CYB_OK;
}

} // extern "C"
