// HBM-bound block-list operations (gfx950): strided N-d copies, BLAS-1 reductions and
// elementwise ops, axis scaling, mask gather/scatter, fills, RNG.
//
// These replace the numpy calls behind the data-movement and BLAS-1 virtuals of the reference's
// NumpyBlockBackend (src/block_backend/numpy.cpp: permute_axes :924-931, reshape :1057-1064,
// apply_mask :605-613, enlarge_leg :700-728, scale_axis :1373-1385, norm :898-913, inner :815-842,
// linear_combination :1358-1365, eye_matrix :1197-1207, zeros :1322-1335) -- one launch per block
// LIST instead of one numpy call per block.  All are bandwidth-class work: 16-B accesses where
// the layout allows, grid-stride loops, deterministic two-stage reductions (no float atomics).
#include "common.h"

#include <algorithm>

namespace {

#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS double* gp;
typedef const GLOBAL_AS double* gcp;

constexpr int NT = 256;
constexpr int64_t CHUNK = 1 << 16; // largest number of elements per workgroup work item
// Work-item size for a list of `total` elements: 64 K elements once the list fills the chip eight workgroups per CU
// deep, smaller (down to 8 K, always a multiple of 1024) for the 10-100 MB lists of one tensor operation, which
// would otherwise run as a few hundred workgroups on 256 CUs.
static int64_t chunk_for(int64_t total)
{
    int64_t c = ((total / 2048) + 1023) & ~(int64_t)1023;
    return std::min(CHUNK, std::max<int64_t>(8192, c));
}

struct Item {
    int32_t desc;
    int32_t pad;
    int64_t start, count;
};

template <typename D>
static int make_items(const D* descs, int64_t n, std::vector<Item>& items, int64_t (*count_of)(const D&))
{
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) total += count_of(descs[i]);
    const int64_t chunk = chunk_for(total);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t tot = count_of(descs[i]);
        for (int64_t s = 0; s < tot; s += chunk) items.push_back(Item{(int32_t)i, 0, s, std::min(chunk, tot - s)});
    }
    return CYB_OK;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------------------------
// strided copy
struct CopyDev {
    void* dst;
    const void* src;
    int32_t ndim, conj;
    int64_t total;
    int64_t shape[CYB_MAX_NDIM];
    int64_t ds[CYB_MAX_NDIM];
    int64_t ss[CYB_MAX_NDIM];
};

typedef unsigned long long u128 __attribute__((ext_vector_type(2)));

// One wave copies one contiguous run of n elements: 16-byte accesses when source and destination are misaligned the
// same way (one element peeled), four independent accesses in flight per lane.
template <typename V>
__device__ __forceinline__ void wave_copy_run(const GLOBAL_AS V* sp, GLOBAL_AS V* dp, int64_t n, int lane, int W = 64)
{
    int64_t i = lane;
    for (; i + 3 * W < n; i += 4 * W) {
        const V a = sp[i], b = sp[i + W], c = sp[i + 2 * W], e = sp[i + 3 * W];
        dp[i] = a;
        dp[i + W] = b;
        dp[i + 2 * W] = c;
        dp[i + 3 * W] = e;
    }
    for (; i < n; i += W) dp[i] = sp[i];
}
// W = 64: the calling wave owns the run; W = NT: the whole workgroup shares it (lane = threadIdx.x)
__device__ __forceinline__ void wave_copy_row8(const GLOBAL_AS uint64_t* sp, GLOBAL_AS uint64_t* dp, int64_t n, int lane, int W = 64)
{
    if (n <= 0) return;
    const unsigned ms = (unsigned)((uintptr_t)sp & 15), md = (unsigned)((uintptr_t)dp & 15);
    if (ms != md) {
        wave_copy_run<uint64_t>(sp, dp, n, lane, W);
        return;
    }
    if (ms) {
        if (lane == 0) dp[0] = sp[0];
        ++sp, ++dp, --n;
    }
    wave_copy_run<u128>((const GLOBAL_AS u128*)sp, (GLOBAL_AS u128*)dp, n >> 1, lane, W);
    if ((n & 1) && lane == 0) dp[n - 1] = sp[n - 1];
}

template <typename T>
__global__ void __launch_bounds__(NT) copy_strided_kernel(const CopyDev* __restrict__ descs, const Item* __restrict__ items)
{
    const Item it = items[blockIdx.x];
    const CopyDev d = descs[it.desc];
    const GLOBAL_AS T* src = (const GLOBAL_AS T*)d.src;
    GLOBAL_AS T* dst = (GLOBAL_AS T*)d.dst;
    // The innermost (merged) axis is contiguous on both sides in most copies (plain copies, the sub-block scatter of
    // combine_legs, the gather of split_legs, permutations that keep the last axis): every wave walks whole rows of it
    // -- no division per element, the outer index is decoded once per row, 16-byte accesses where the alignment allows.
    const int last = d.ndim - 1;
    if (d.ndim >= 1 && d.ss[last] == 1 && d.ds[last] == 1 && d.shape[last] >= 16 && !(sizeof(T) == 16 && d.conj)) {
        const int64_t inner = d.shape[last];
        const int64_t e0 = it.start, e1 = it.start + it.count;
        const int64_t r0 = e0 / inner, r1 = (e1 - 1) / inner;
        // short rows: one wave per row (four rows in flight per workgroup); long rows: the waves share a row
        const bool shared_row = inner >= 2048;
        const int wave = shared_row ? 0 : (int)(threadIdx.x >> 6), lane = shared_row ? (int)threadIdx.x : (int)(threadIdx.x & 63);
        for (int64_t row = r0 + wave; row <= r1; row += shared_row ? 1 : NT / 64) {
            const int64_t c0 = (row == r0) ? e0 - r0 * inner : 0;
            const int64_t c1 = (row == r1) ? e1 - r1 * inner : inner;
            int64_t rem = row, so = 0, dof = 0;
            for (int k = last - 1; k >= 0; --k) {
                const int64_t sh = d.shape[k];
                const int64_t q = rem / sh, i = rem - q * sh;
                rem = q;
                so += i * d.ss[k];
                dof += i * d.ds[k];
            }
            const GLOBAL_AS T* sp = src + so + c0;
            GLOBAL_AS T* dp = dst + dof + c0;
            const int W = shared_row ? NT : 64;
            if constexpr (sizeof(T) == 8) wave_copy_row8((const GLOBAL_AS uint64_t*)sp, (GLOBAL_AS uint64_t*)dp, c1 - c0, lane, W);
            else wave_copy_run<T>(sp, dp, c1 - c0, lane, W);
        }
        return;
    }
    for (int64_t e = it.start + threadIdx.x; e < it.start + it.count; e += NT) {
        int64_t rem = e, so = 0, dof = 0;
#pragma unroll
        for (int k = CYB_MAX_NDIM - 1; k >= 0; --k) {
            if (k < d.ndim) {
                const int64_t sh = d.shape[k];
                const int64_t q = rem / sh, i = rem - q * sh;
                rem = q;
                so += i * d.ss[k];
                dof += i * d.ds[k];
            }
        }
        T v = src[so];
        if constexpr (sizeof(T) == 16) {
            if (d.conj) v.y ^= 0x8000000000000000ull; // flip the sign of the imaginary part
        }
        dst[dof] = v;
    }
}


// Transposing copies (the fastest axis of the destination is not the fastest axis of the source: permute_axes
// of a compose operand, the leg rotations of a Krylov matvec): 32 x 32 tiles through LDS so that BOTH the reads
// (along the source's unit-stride axis S) and the writes (along the destination's unit-stride axis D) are
// coalesced, and the index arithmetic (64-bit div/mod over up to 8 axes) runs once per tile, not per element.
struct CopyT {
    void* dst;
    const void* src;
    int32_t n_outer, conj;
    int64_t nS, nD;       // extents of the two tiled axes
    int64_t ssD, dsS;     // source stride of D, destination stride of S (ss of S and ds of D are 1)
    // A tiled axis may be the flattening of TWO axes that are contiguous on its own side (a short innermost axis
    // such as the MPO bond of [.., vR, wR] and its neighbour): index i of D then sits at source offset
    // (i / nD2) * ssD + (i % nD2) * ssD2, index i of S at destination offset (i / nS2) * dsS + (i % nS2) * dsS2.
    // nD2 = nS2 = 1 for a plain axis.
    int64_t nD2, ssD2, nS2, dsS2;
    int64_t tilesS, tilesD;
    int64_t oshape[CYB_MAX_NDIM], ods[CYB_MAX_NDIM], oss[CYB_MAX_NDIM]; // the remaining (outer) axes
};

template <typename T>
__global__ void __launch_bounds__(NT) copy_transpose_kernel(const CopyT* __restrict__ descs, const Item* __restrict__ items)
{
    __shared__ T tile[32][33];
    const Item it = items[blockIdx.x];
    const CopyT d = descs[it.desc];
    const GLOBAL_AS T* src = (const GLOBAL_AS T*)d.src;
    GLOBAL_AS T* dst = (GLOBAL_AS T*)d.dst;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int64_t t = it.start; t < it.start + it.count; ++t) {
        int64_t rem = t;
        const int64_t td = rem % d.tilesD;
        rem /= d.tilesD;
        const int64_t ts = rem % d.tilesS;
        rem /= d.tilesS;
        int64_t so = 0, dof = 0;
        for (int k = d.n_outer - 1; k >= 0; --k) {
            const int64_t q = rem / d.oshape[k], i = rem - q * d.oshape[k];
            rem = q;
            so += i * d.oss[k];
            dof += i * d.ods[k];
        }
        const int64_t s0 = ts * 32, d0 = td * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t sI = s0 + tx, dI = d0 + ty + 8 * q;
            if (sI < d.nS && dI < d.nD) {
                T v = src[so + sI + (dI / d.nD2) * d.ssD + (dI % d.nD2) * d.ssD2];
                if constexpr (sizeof(T) == 16) {
                    if (d.conj) v.y ^= 0x8000000000000000ull;
                }
                tile[ty + 8 * q][tx] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t dI = d0 + tx, sI = s0 + ty + 8 * q;
            if (sI < d.nS && dI < d.nD) dst[dof + dI + (sI / d.nS2) * d.dsS + (sI % d.nS2) * d.dsS2] = tile[tx][ty + 8 * q];
        }
        __syncthreads();
    }
}

// 8-byte elements: 64 x 64 tiles, two elements per 16-byte access on both sides (falls back to 8-byte accesses
// row by row when a row start is not 16-byte aligned).  LDS image is [s][d] so that the write phase reads pairs.
__global__ void __launch_bounds__(NT) copy_transpose64_kernel(const CopyT* __restrict__ descs, const Item* __restrict__ items)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    constexpr int TS = 64, LS = TS + 2;
    __shared__ __attribute__((aligned(16))) double tile[TS * LS]; // tile[s * LS + d]
    const Item it = items[blockIdx.x];
    const CopyT d = descs[it.desc];
    gcp src = (gcp)d.src;
    gp dst = (gp)d.dst;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int64_t t = it.start; t < it.start + it.count; ++t) {
        int64_t rem = t;
        const int64_t td = rem % d.tilesD;
        rem /= d.tilesD;
        const int64_t ts = rem % d.tilesS;
        rem /= d.tilesS;
        int64_t so = 0, dof = 0;
        for (int k = d.n_outer - 1; k >= 0; --k) {
            const int64_t q = rem / d.oshape[k], i = rem - q * d.oshape[k];
            rem = q;
            so += i * d.oss[k];
            dof += i * d.ods[k];
        }
        const int64_t s0 = ts * TS, d0 = td * TS;
#pragma unroll
        for (int q = 0; q < 8; ++q) { // read: rows along D, pairs along S
            const int64_t dI = d0 + ty + 8 * q, sI = s0 + 2 * tx;
            if (dI < d.nD && sI < d.nS) {
                gcp p = src + so + (dI / d.nD2) * d.ssD + (dI % d.nD2) * d.ssD2 + sI;
                double v0, v1 = 0.0;
                if (sI + 1 < d.nS && (((uintptr_t)p) & 15) == 0) {
                    const d2v v = *(const GLOBAL_AS d2v*)p;
                    v0 = v.x;
                    v1 = v.y;
                } else {
                    v0 = p[0];
                    if (sI + 1 < d.nS) v1 = p[1];
                }
                tile[(2 * tx) * LS + ty + 8 * q] = v0;
                tile[(2 * tx + 1) * LS + ty + 8 * q] = v1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) { // write: rows along S, pairs along D
            const int64_t sI = s0 + ty + 8 * q, dI = d0 + 2 * tx;
            if (sI < d.nS && dI < d.nD) {
                gp p = dst + dof + (sI / d.nS2) * d.dsS + (sI % d.nS2) * d.dsS2 + dI;
                const d2v v = *reinterpret_cast<const d2v*>(&tile[(ty + 8 * q) * LS + 2 * tx]);
                if (dI + 1 < d.nD && (((uintptr_t)p) & 15) == 0) {
                    *(GLOBAL_AS d2v*)p = v;
                } else {
                    p[0] = v.x;
                    if (dI + 1 < d.nD) p[1] = v.y;
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// complex128 support of the tdot path.  A complex block is stored interleaved (re, im) like numpy's
// complex128.  A complex product C = A B runs through the SAME real f64-MFMA grouped GEMM: A (M x K complex,
// k-contiguous) is read in place as the real M x 2K matrix [ar0 ai0 ar1 ai1 ...], C (M x N complex) is written in
// place as the real M x 2N matrix, and B is expanded once into the real 2K x 2N matrix
//        B'[2k][2n] = br   B'[2k][2n+1] = bi   B'[2k+1][2n] = -bi   B'[2k+1][2n+1] = br
// so that A_real B' = C_real.  8 M N K real flops -- exactly the four real products of a complex one.
struct CExpandDev {
    const double* src; // complex elements (re, im)
    int64_t rs, cs;    // strides of src in complex elements
    int64_t K, N;
    double* dst;       // 2K x 2N doubles, row-major, contiguous
};

__global__ void __launch_bounds__(NT) complex_expand_kernel(const CExpandDev* __restrict__ descs, const Item* __restrict__ items)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    const Item it = items[blockIdx.x];
    const CExpandDev d = descs[it.desc];
    const GLOBAL_AS d2v* src = (const GLOBAL_AS d2v*)d.src;
    gp dst = (gp)d.dst;
    for (int64_t e = it.start + threadIdx.x; e < it.start + it.count; e += NT) {
        const int64_t k = e / d.N, n = e - k * d.N;
        const d2v b = src[k * d.rs + n * d.cs];
        GLOBAL_AS d2v* r0 = (GLOBAL_AS d2v*)(dst + (2 * k) * (2 * d.N) + 2 * n);
        GLOBAL_AS d2v* r1 = (GLOBAL_AS d2v*)(dst + (2 * k + 1) * (2 * d.N) + 2 * n);
        *r0 = d2v{b.x, b.y};
        *r1 = d2v{-b.y, b.x};
    }
}

// ---------------------------------------------------------------------------------------------
// reductions (two-stage, deterministic)
struct VecDev {
    const double* x;
    const double* y;
    double* out;
    int64_t n;
};

// mode 0: sum x*y (y null: x*x) ; mode 1: max |x|
__device__ __forceinline__ void reduce_stage1_body(const VecDev& d, const Item& it, double* __restrict__ partial, int mode)
{
    __shared__ double red[NT / 64];
    gcp x = (gcp)d.x;
    gcp y = (gcp)d.y;
    double acc = 0.0;
    const int64_t e1 = it.start + it.count;
    // 16-byte accesses when the operands allow (chunk starts are even): half the memory instructions per byte
    typedef double d2v __attribute__((ext_vector_type(2)));
    const bool al = ((((uintptr_t)(x + it.start)) | (d.y ? (uintptr_t)(y + it.start) : 0)) & 15) == 0;
    const int64_t nv = al ? it.count / 2 : 0;
    const GLOBAL_AS d2v* xv = (const GLOBAL_AS d2v*)(x + it.start);
    const GLOBAL_AS d2v* yv = (const GLOBAL_AS d2v*)(y + it.start);
    if (mode == 0) {
        double acc2 = 0.0;
        if (d.y) {
            for (int64_t i = threadIdx.x; i < nv; i += NT) {
                const d2v a = xv[i], b = yv[i];
                acc += a.x * b.x;
                acc2 += a.y * b.y;
            }
            for (int64_t e = it.start + 2 * nv + threadIdx.x; e < e1; e += NT) acc += x[e] * y[e];
        } else {
            for (int64_t i = threadIdx.x; i < nv; i += NT) {
                const d2v a = xv[i];
                acc += a.x * a.x;
                acc2 += a.y * a.y;
            }
            for (int64_t e = it.start + 2 * nv + threadIdx.x; e < e1; e += NT) acc += x[e] * x[e];
        }
        acc = wave_sum(acc + acc2);
    } else {
        for (int64_t i = threadIdx.x; i < nv; i += NT) {
            const d2v a = xv[i];
            acc = fmax(acc, fmax(fabs(a.x), fabs(a.y)));
        }
        for (int64_t e = it.start + 2 * nv + threadIdx.x; e < e1; e += NT) acc = fmax(acc, fabs(x[e]));
        acc = wave_max(acc);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int q = 1; q < NT / 64; ++q) r = (mode == 0) ? r + red[q] : fmax(r, red[q]);
        partial[blockIdx.x] = r;
    }
}
__global__ void __launch_bounds__(NT) reduce_stage1_kernel(const VecDev* __restrict__ descs, const Item* __restrict__ items,
                                                           double* __restrict__ partial, int mode)
{
    const Item it = items[blockIdx.x];
    const VecDev d = descs[it.desc];
    reduce_stage1_body(d, it, partial, mode);
}
// ONE vector: descriptor and chunking travel as kernel arguments -- no descriptor upload (a host call and a copy kernel per
// launch: most BLAS-1 calls of a Krylov iteration on flat pools are of this kind)
__global__ void __launch_bounds__(NT) reduce_stage1_one_kernel(VecDev d, int64_t chunk, double* __restrict__ partial, int mode)
{
    const int64_t start = (int64_t)blockIdx.x * chunk;
    const Item it{0, 0, start, min(chunk, d.n - start)};
    reduce_stage1_body(d, it, partial, mode);
}

// result[g] = reduce over partial[seg[g] .. seg[g+1])  (one workgroup per group, fixed order)
__device__ __forceinline__ void reduce_stage2_body(const double* __restrict__ partial, int64_t s0, int64_t s1, double* __restrict__ result, int mode)
{
    __shared__ double red[NT / 64];
    double acc = 0.0;
    for (int64_t e = s0 + threadIdx.x; e < s1; e += NT) acc = (mode == 0) ? acc + partial[e] : fmax(acc, partial[e]);
    acc = (mode == 0) ? wave_sum(acc) : wave_max(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int q = 1; q < NT / 64; ++q) r = (mode == 0) ? r + red[q] : fmax(r, red[q]);
        result[blockIdx.x] = r;
    }
}
__global__ void __launch_bounds__(NT) reduce_stage2_kernel(const double* __restrict__ partial, const int64_t* __restrict__ seg,
                                                           double* __restrict__ result, int mode)
{
    reduce_stage2_body(partial, seg[blockIdx.x], seg[blockIdx.x + 1], result, mode);
}
__global__ void __launch_bounds__(NT) reduce_stage2_one_kernel(const double* __restrict__ partial, int64_t n_items, double* __restrict__ result, int mode)
{
    reduce_stage2_body(partial, 0, n_items, result, mode);
}

// out = a*x + b*y on complex vectors (a, b complex scalars; y may be null)
__global__ void __launch_bounds__(NT) axpby_c128_kernel(const VecDev* __restrict__ descs, const Item* __restrict__ items,
                                                        double ar, double ai, double br, double bi)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    const Item it = items[blockIdx.x];
    const VecDev d = descs[it.desc];
    const GLOBAL_AS d2v* x = (const GLOBAL_AS d2v*)d.x;
    const GLOBAL_AS d2v* y = (const GLOBAL_AS d2v*)d.y;
    GLOBAL_AS d2v* out = (GLOBAL_AS d2v*)d.out;
    for (int64_t e = it.start + threadIdx.x; e < it.start + it.count; e += NT) {
        const d2v xv = x[e];
        d2v r = d2v{ar * xv.x - ai * xv.y, ar * xv.y + ai * xv.x};
        if (d.y) {
            const d2v yv = y[e];
            r.x += br * yv.x - bi * yv.y;
            r.y += br * yv.y + bi * yv.x;
        }
        out[e] = r;
    }
}

// complex elementwise: op 0 |z| (real out), 1 sqrt, 2 exp, 3 log, 4 angle (real out), 5 z*w, 6 z/w (Smith)
__global__ void __launch_bounds__(NT) celementwise_kernel(const VecDev* __restrict__ descs, const Item* __restrict__ items, int op)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    const Item it = items[blockIdx.x];
    const VecDev d = descs[it.desc];
    const GLOBAL_AS d2v* x = (const GLOBAL_AS d2v*)d.x;
    const GLOBAL_AS d2v* y = (const GLOBAL_AS d2v*)d.y;
    GLOBAL_AS d2v* outc = (GLOBAL_AS d2v*)d.out;
    gp outr = (gp)d.out;
    for (int64_t e = it.start + threadIdx.x; e < it.start + it.count; e += NT) {
        const d2v z = x[e];
        if (op == 0) {
            outr[e] = hypot(z.x, z.y);
            continue;
        }
        if (op == 4) {
            outr[e] = atan2(z.y, z.x);
            continue;
        }
        d2v r;
        switch (op) {
        case 1: { // principal square root
            const double m = hypot(z.x, z.y);
            if (m == 0.0) {
                r = d2v{0.0, z.y};
            } else {
                const double t = sqrt(0.5 * (m + fabs(z.x)));
                r = z.x >= 0.0 ? d2v{t, z.y / (2.0 * t)} : d2v{fabs(z.y) / (2.0 * t), copysign(t, z.y)};
            }
            break;
        }
        case 2: {
            const double ex = exp(z.x);
            double sn, cs;
            sincos(z.y, &sn, &cs);
            r = d2v{ex * cs, ex * sn};
            break;
        }
        case 3: r = d2v{log(hypot(z.x, z.y)), atan2(z.y, z.x)}; break;
        case 5: {
            const d2v w = y[e];
            r = d2v{z.x * w.x - z.y * w.y, z.x * w.y + z.y * w.x};
            break;
        }
        default: {
            const d2v w = y[e];
            if (fabs(w.x) >= fabs(w.y)) {
                const double q = w.y / w.x, den = w.x + w.y * q;
                r = d2v{(z.x + z.y * q) / den, (z.y - z.x * q) / den};
            } else {
                const double q = w.x / w.y, den = w.x * q + w.y;
                r = d2v{(z.x * q + z.y) / den, (z.y * q - z.x) / den};
            }
            break;
        }
        }
        outc[e] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// elementwise
// kind 0: out = a*x + b*y (y may be null) ; kind 1: binary op ; kind 2: unary op ; kind 3: unary op with parameter a
// numpy's ``x ** y`` is exact whenever the result is representable (4.0 ** 3.0 == 64.0 is asserted by the reference's own
// test, test_block_backend_cpp.py:127); the device library's pow() is only 1-ulp accurate.  Integer exponents of moderate
// size go through exponentiation by squaring (every partial product exact in the representable cases).
__device__ __forceinline__ double pow_exactish(double x, double y)
{
    const double yi = rint(y);
    if (yi == y && fabs(y) <= 4096.0) {
        unsigned int n = (unsigned int)fabs(yi);
        double base = x, r = 1.0;
        while (n) {
            if (n & 1u) r *= base;
            base *= base;
            n >>= 1;
        }
        return yi < 0.0 ? 1.0 / r : r;
    }
    return pow(x, y);
}

__device__ __forceinline__ void elementwise_body(const VecDev& d, const Item& it, int kind, int op, double a, double b)
{
    gcp x = (gcp)d.x;
    gcp y = (gcp)d.y;
    gp out = (gp)d.out;
    const int64_t e1 = it.start + it.count;
    if (kind == 0) { // a*x + b*y (the Krylov axpy / scale): 16-byte accesses when the three operands allow
        typedef double d2v __attribute__((ext_vector_type(2)));
        const bool al = ((((uintptr_t)(x + it.start)) | ((uintptr_t)(out + it.start)) | (d.y ? (uintptr_t)(y + it.start) : 0)) & 15) == 0;
        const int64_t nv = al ? it.count / 2 : 0;
        const GLOBAL_AS d2v* xv2 = (const GLOBAL_AS d2v*)(x + it.start);
        const GLOBAL_AS d2v* yv2 = (const GLOBAL_AS d2v*)(y + it.start);
        GLOBAL_AS d2v* ov2 = (GLOBAL_AS d2v*)(out + it.start);
        if (d.y) {
            for (int64_t i = threadIdx.x; i < nv; i += NT) {
                const d2v xa = xv2[i], ya = yv2[i];
                ov2[i] = d2v{a * xa.x + b * ya.x, a * xa.y + b * ya.y};
            }
            for (int64_t e = it.start + 2 * nv + threadIdx.x; e < e1; e += NT) out[e] = a * x[e] + b * y[e];
        } else {
            for (int64_t i = threadIdx.x; i < nv; i += NT) {
                const d2v xa = xv2[i];
                ov2[i] = d2v{a * xa.x, a * xa.y};
            }
            for (int64_t e = it.start + 2 * nv + threadIdx.x; e < e1; e += NT) out[e] = a * x[e];
        }
        return;
    }
    for (int64_t e = it.start + threadIdx.x; e < e1; e += NT) {
        const double xv = x[e];
        double r;
        if (kind == 0) {
            r = d.y ? a * xv + b * y[e] : a * xv;
        } else if (kind == 1) {
            const double yv = y[e];
            r = op == 0 ? xv + yv : op == 1 ? xv - yv : op == 2 ? xv * yv : op == 3 ? xv / yv : pow_exactish(xv, yv);
        } else if (kind == 3) { // unary with one parameter `a`
            switch (op) {
            case 0: r = fabs(xv) < a ? 0.0 : 1.0 / xv; break;        // cutoff_inverse
            case 1: r = xv > a ? log(xv) : 0.0; break;                // stable_log
            case 2: r = pow_exactish(xv, a); break;                   // Block::pow(Scalar)
            default: r = xv != xv ? xv : (signbit(xv) ? 3.141592653589793 : 0.0); break; // angle of a real number
            }
        } else {
            switch (op) {
            case 0: r = fabs(xv); break;
            case 1: r = sqrt(xv); break;
            case 2: r = exp(xv); break;
            case 3: r = log(xv); break;
            case 4: r = -xv; break;
            case 5: r = xv * xv; break;
            case 6: r = 1.0 / xv; break;
            case 7: r = (double)(float)xv; break;       // round to float32: the cast-on-store of float32 / complex64 blocks
            default: r = trunc(xv); break;               // int64 blocks hold integers (np.asarray(x, int64) truncates)
            }
        }
        out[e] = r;
    }
}
__global__ void __launch_bounds__(NT) elementwise_kernel(const VecDev* __restrict__ descs, const Item* __restrict__ items,
                                                         int kind, int op, double a, double b)
{
    const Item it = items[blockIdx.x];
    const VecDev d = descs[it.desc];
    elementwise_body(d, it, kind, op, a, b);
}
__global__ void __launch_bounds__(NT) elementwise_one_kernel(VecDev d, int64_t chunk, int kind, int op, double a, double b)
{   // (ONE vector: see reduce_stage1_one_kernel)
    const int64_t start = (int64_t)blockIdx.x * chunk;
    const Item it{0, 0, start, min(chunk, d.n - start)};
    elementwise_body(d, it, kind, op, a, b);
}

struct ScaleDev {
    const double* x;
    const double* f;
    double* out;
    int64_t outer, axis, inner;
};
__global__ void __launch_bounds__(NT) scale_axis_kernel(const ScaleDev* __restrict__ descs, const Item* __restrict__ items)
{
    const Item it = items[blockIdx.x];
    const ScaleDev d = descs[it.desc];
    gcp x = (gcp)d.x;
    gcp f = (gcp)d.f;
    gp out = (gp)d.out;
    const int64_t e1 = it.start + it.count;
    // pairs of elements per thread (16-byte accesses) when both buffers are 16-byte aligned; the factor index of
    // the second element of a pair is derived from the first one's without a second division
    typedef double d2v __attribute__((ext_vector_type(2)));
    const bool al = ((((uintptr_t)(x + it.start)) | ((uintptr_t)(out + it.start))) & 15) == 0;
    const int64_t nv = al ? it.count / 2 : 0;
    const GLOBAL_AS d2v* xv = (const GLOBAL_AS d2v*)(x + it.start);
    GLOBAL_AS d2v* ov = (GLOBAL_AS d2v*)(out + it.start);
    for (int64_t i = threadIdx.x; i < nv; i += NT) {
        const int64_t e = it.start + 2 * i;
        const int64_t t = e / d.inner, in = e - t * d.inner;
        const int64_t j0 = t % d.axis;
        int64_t j1 = j0;
        if (in + 1 == d.inner) j1 = (j0 + 1 == d.axis) ? 0 : j0 + 1;
        const d2v v = xv[i];
        ov[i] = d2v{v.x * f[j0], v.y * f[j1]};
    }
    for (int64_t e = it.start + 2 * nv + threadIdx.x; e < e1; e += NT) {
        const int64_t j = (e / d.inner) % d.axis;
        out[e] = x[e] * f[j];
    }
}

struct MaskDev {
    const double* x;
    double* out;
    const int64_t* idx;
    int64_t outer, axis, inner, n_keep;
};
// work items enumerate the elements of the SMALL side (outer, n_keep, inner)
__global__ void __launch_bounds__(NT) mask_kernel(const MaskDev* __restrict__ descs, const Item* __restrict__ items, int scatter)
{
    const Item it = items[blockIdx.x];
    const MaskDev d = descs[it.desc];
    gcp x = (gcp)d.x;
    gp out = (gp)d.out;
    const GLOBAL_AS int64_t* idx = (const GLOBAL_AS int64_t*)d.idx;
    const int64_t e1 = it.start + it.count;
    if (d.inner >= 16) {
        // row gather / scatter (the mask acts on a leading axis: Vh of a truncated SVD): every wave moves whole kept
        // rows, the kept index is decoded once per row
        const int64_t inner = d.inner;
        const int64_t r0 = it.start / inner, r1 = (e1 - 1) / inner;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int64_t row = r0 + wave; row <= r1; row += NT / 64) {
            const int64_t c0 = (row == r0) ? it.start - r0 * inner : 0;
            const int64_t c1 = (row == r1) ? e1 - r1 * inner : inner;
            const int64_t o = row / d.n_keep, j = row - o * d.n_keep;
            const int64_t big = (o * d.axis + idx[j]) * inner + c0, sm = row * inner + c0;
            const GLOBAL_AS uint64_t* sp = (const GLOBAL_AS uint64_t*)(x + (scatter ? sm : big));
            GLOBAL_AS uint64_t* dp = (GLOBAL_AS uint64_t*)(out + (scatter ? big : sm));
            wave_copy_row8(sp, dp, c1 - c0, lane);
        }
        return;
    }
    if (d.outer * d.axis * d.inner < ((int64_t)1 << 31)) {
        // short inner runs (inner = 1: the mask acts on the last axis, U of a truncated SVD): 32-bit index arithmetic
        const uint32_t inner = (uint32_t)d.inner, nk = (uint32_t)d.n_keep, axis = (uint32_t)d.axis;
        for (uint32_t e = (uint32_t)it.start + threadIdx.x; e < (uint32_t)e1; e += NT) {
            const uint32_t t = e / inner, in = e - t * inner;
            const uint32_t o = t / nk, j = t - o * nk;
            const uint32_t big = (o * axis + (uint32_t)idx[j]) * inner + in;
            if (scatter) out[big] = x[e];
            else out[e] = x[big];
        }
        return;
    }
    if ((d.inner & 1) == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0) {
        // even inner extent: a pair of neighbours never straddles a kept slice, 16-byte accesses on both sides
        typedef double d2v __attribute__((ext_vector_type(2)));
        for (int64_t e = it.start + 2 * threadIdx.x; e < e1; e += 2 * NT) {
            const int64_t t = e / d.inner, in = e - t * d.inner;
            const int64_t j = t % d.n_keep, o = t / d.n_keep;
            const int64_t big = (o * d.axis + idx[j]) * d.inner + in;
            if (scatter) *(GLOBAL_AS d2v*)(out + big) = *(const GLOBAL_AS d2v*)(x + e);
            else *(GLOBAL_AS d2v*)(out + e) = *(const GLOBAL_AS d2v*)(x + big);
        }
        return;
    }
    for (int64_t e = it.start + threadIdx.x; e < e1; e += NT) {
        const int64_t in = e % d.inner;
        const int64_t t = e / d.inner;
        const int64_t j = t % d.n_keep, o = t / d.n_keep;
        const int64_t big = (o * d.axis + idx[j]) * d.inner + in;
        if (scatter) out[big] = x[e];
        else out[e] = x[big];
    }
}

__global__ void __launch_bounds__(NT) fill_kernel(double* __restrict__ out, int64_t n, double v)
{
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) out[e] = v;
}
__global__ void __launch_bounds__(NT) eye_kernel(double* __restrict__ out, int64_t n)
{
    const int64_t tot = n * n;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < tot; e += (int64_t)gridDim.x * NT)
        out[e] = (e / n == e % n) ? 1.0 : 0.0;
}

// Philox4x32-10 counter-based generator + Box-Muller
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
}
__global__ void __launch_bounds__(NT) random_normal_kernel(double* __restrict__ out, int64_t n, uint64_t seed, double sigma)
{
    const int64_t npair = (n + 1) / 2;
    for (int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x; p < npair; p += (int64_t)gridDim.x * NT) {
        uint32_t c[4] = {(uint32_t)p, (uint32_t)((uint64_t)p >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const uint64_t a = ((uint64_t)c[0] << 32) | c[1];
        const uint64_t b = ((uint64_t)c[2] << 32) | c[3];
        const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740992.0); // (0,1]
        const double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);         // [0,1)
        const double rad = sigma * sqrt(-2.0 * log(u1));
        double s, co;
        sincos(6.283185307179586 * u2, &s, &co);
        out[2 * p] = rad * co;
        if (2 * p + 1 < n) out[2 * p + 1] = rad * s;
    }
}

__global__ void __launch_bounds__(NT) random_uniform_kernel(double* __restrict__ out, int64_t n, uint64_t seed, double lo, double hi)
{
    const int64_t npair = (n + 1) / 2;
    for (int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x; p < npair; p += (int64_t)gridDim.x * NT) {
        uint32_t c[4] = {(uint32_t)p, (uint32_t)((uint64_t)p >> 32), 1u, 0u}; // counter word 2 separates the uniform stream
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const uint64_t a = ((uint64_t)c[0] << 32) | c[1];
        const uint64_t b = ((uint64_t)c[2] << 32) | c[3];
        out[2 * p] = lo + (hi - lo) * ((double)(a >> 11) * (1.0 / 9007199254740992.0));
        if (2 * p + 1 < n) out[2 * p + 1] = lo + (hi - lo) * ((double)(b >> 11) * (1.0 / 9007199254740992.0));
    }
}

// ---------------------------------------------------------------------------------------------
// comparisons -> boolean (one byte per element) blocks, and their reductions
__global__ void __launch_bounds__(NT) compare_kernel(const double* __restrict__ x_, const double* __restrict__ y_, double scalar,
                                                     uint8_t* __restrict__ out_, int64_t n, int op)
{
    gcp x = (gcp)x_;
    gcp y = (gcp)y_;
    GLOBAL_AS uint8_t* out = (GLOBAL_AS uint8_t*)out_;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) {
        const double a = x[e], b = y ? y[e] : scalar;
        bool r;
        switch (op) {
        case 0: r = a < b; break;
        case 1: r = a <= b; break;
        case 2: r = a > b; break;
        case 3: r = a >= b; break;
        case 4: r = a == b; break;
        default: r = a != b; break;
        }
        out[e] = r ? 1 : 0;
    }
}
__global__ void __launch_bounds__(NT) convert_u8_f64_kernel(const uint8_t* __restrict__ x_, double* __restrict__ out_, int64_t n)
{
    const GLOBAL_AS uint8_t* x = (const GLOBAL_AS uint8_t*)x_;
    gp out = (gp)out_;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) out[e] = x[e] ? 1.0 : 0.0;
}
__global__ void __launch_bounds__(NT) count_nonzero_kernel(const uint8_t* __restrict__ x_, int64_t n, unsigned long long* __restrict__ result)
{
    const GLOBAL_AS uint8_t* x = (const GLOBAL_AS uint8_t*)x_;
    unsigned long long c = 0;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) c += x[e] != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(result, c); // integer atomics: order-independent, deterministic
}

// extremum with its flat index (ties: lowest index, like np.argmax / np.argmin): mode 0 max, 1 min, 2 max |x|
struct Ext {
    double v;
    long long i;
};
__device__ __forceinline__ bool ext_better(double v, long long i, double bv, long long bi)
{
    return bi < 0 || v > bv || (v == bv && i < bi);
}
__device__ __forceinline__ void ext_wave(double& v, long long& i)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const long long oi = __shfl_xor(i, o);
        if (oi >= 0 && ext_better(ov, oi, v, i)) v = ov, i = oi;
    }
}
__global__ void __launch_bounds__(NT) extremum_kernel(const double* __restrict__ x_, int64_t n, int mode, Ext* __restrict__ partial,
                                                      const Ext* __restrict__ prev, int64_t n_prev)
{
    __shared__ double sv[NT / 64];
    __shared__ long long si[NT / 64];
    gcp x = (gcp)x_;
    double bv = 0.0;
    long long bi = -1;
    if (prev) { // second stage: reduce the partial records
        for (int64_t e = threadIdx.x; e < n_prev; e += NT)
            if (prev[e].i >= 0 && ext_better(prev[e].v, prev[e].i, bv, bi)) bv = prev[e].v, bi = prev[e].i;
    } else {
        for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) {
            const double r = x[e];
            const double key = mode == 0 ? r : mode == 1 ? -r : fabs(r); // every mode is a maximisation of `key`
            if (ext_better(key, e, bv, bi)) bv = key, bi = e;
        }
    }
    ext_wave(bv, bi);
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = bv, si[threadIdx.x >> 6] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < NT / 64; ++q)
            if (si[q] >= 0 && ext_better(sv[q], si[q], bv, bi)) bv = sv[q], bi = si[q];
        partial[blockIdx.x] = Ext{bv, bi};
    }
}

// ---------------------------------------------------------------------------------------------
// dst[idx] = (accumulate ? dst[idx] : 0) + sum_t coeff_t * src_t[idx] over N-d strided views: the tree-block updates of
// FusionTreeBackend::apply_instructions (TreePairMapping::transform_tensor, fusion_tree_mapping.cpp:391-513):
// `tree_block = sum_I f_JI * old_block[slices_I]`, `permute_combined_matrix`, `new_block[slices] = ...` as ONE
// launch per tensor -- the permutation is in the source strides, the sub-block placement in the destination view.
struct LinDev {
    double* dst;
    int32_t ndim, accumulate, term_begin, term_end;
    int64_t total;
    int64_t shape[CYB_MAX_NDIM], ds[CYB_MAX_NDIM];
};
struct LinTerm {
    const double* src;
    double coeff;
    int64_t ss[CYB_MAX_NDIM];
};
__global__ void __launch_bounds__(NT) lincomb_strided_kernel(const LinDev* __restrict__ descs, const LinTerm* __restrict__ terms,
                                                             const Item* __restrict__ items)
{
    const Item it = items[blockIdx.x];
    const LinDev d = descs[it.desc];
    gp dst = (gp)d.dst;
    for (int64_t e = it.start + threadIdx.x; e < it.start + it.count; e += NT) {
        int64_t idx[CYB_MAX_NDIM];
        int64_t rem = e, dof = 0;
#pragma unroll
        for (int k = CYB_MAX_NDIM - 1; k >= 0; --k) {
            idx[k] = 0;
            if (k < d.ndim) {
                const int64_t q = rem / d.shape[k];
                idx[k] = rem - q * d.shape[k];
                rem = q;
                dof += idx[k] * d.ds[k];
            }
        }
        double acc = d.accumulate ? dst[dof] : 0.0;
        for (int t = d.term_begin; t < d.term_end; ++t) {
            const GLOBAL_AS LinTerm* tm = (const GLOBAL_AS LinTerm*)(terms + t);
            int64_t so = 0;
#pragma unroll
            for (int k = 0; k < CYB_MAX_NDIM; ++k)
                if (k < d.ndim) so += idx[k] * tm->ss[k];
            acc += tm->coeff * ((gcp)tm->src)[so];
        }
        dst[dof] = acc;
    }
}

// complex128 form: dst and (unless src_real) the sources are interleaved (re, im) arrays, strides in complex elements; a
// term with src_real reads a float64 array (strides in doubles) -- real data under a complex mapping (anyonic R / C
// symbols): `dtype = to_complex(dtype)` at fusion_tree_mapping.cpp:433-436
struct LinTermC {
    const double* src;
    double cr, ci;
    int32_t src_real, pad;
    int64_t ss[CYB_MAX_NDIM];
};
__global__ void __launch_bounds__(NT) lincomb_strided_c128_kernel(const LinDev* __restrict__ descs, const LinTermC* __restrict__ terms,
                                                                  const Item* __restrict__ items)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const Item it = items[blockIdx.x];
    const LinDev d = descs[it.desc];
    GLOBAL_AS d2* dst = (GLOBAL_AS d2*)d.dst;
    for (int64_t e = it.start + threadIdx.x; e < it.start + it.count; e += NT) {
        int64_t idx[CYB_MAX_NDIM];
        int64_t rem = e, dof = 0;
#pragma unroll
        for (int k = CYB_MAX_NDIM - 1; k >= 0; --k) {
            idx[k] = 0;
            if (k < d.ndim) {
                const int64_t q = rem / d.shape[k];
                idx[k] = rem - q * d.shape[k];
                rem = q;
                dof += idx[k] * d.ds[k];
            }
        }
        d2 acc = d.accumulate ? dst[dof] : d2{0.0, 0.0};
        for (int t = d.term_begin; t < d.term_end; ++t) {
            const GLOBAL_AS LinTermC* tm = (const GLOBAL_AS LinTermC*)(terms + t);
            int64_t so = 0;
#pragma unroll
            for (int k = 0; k < CYB_MAX_NDIM; ++k)
                if (k < d.ndim) so += idx[k] * tm->ss[k];
            d2 x;
            if (tm->src_real) x = d2{((gcp)tm->src)[so], 0.0};
            else x = ((const GLOBAL_AS d2*)tm->src)[so];
            acc.x += tm->cr * x.x - tm->ci * x.y;
            acc.y += tm->cr * x.y + tm->ci * x.x;
        }
        dst[dof] = acc;
    }
}

static int64_t vec_count(const cyb_vec_desc& d) { return d.n; }
static int64_t scale_count(const cyb_scale_axis_desc& d) { return d.outer * d.axis * d.inner; }
static int64_t mask_count(const cyb_mask_desc& d) { return d.outer * d.n_keep * d.inner; }

static int upload_vecs(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, bool need_y, bool need_out,
                       std::vector<Item>& items, void** d_descs, void** d_items, bool defer = false, std::vector<VecDev>* keep = nullptr)
{
    std::vector<VecDev> hv((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        CYB_REQUIRE(descs[i].n >= 0, "vector desc %lld: negative length", (long long)i);
        CYB_REQUIRE(descs[i].n == 0 || descs[i].x, "vector desc %lld: x is NULL", (long long)i);
        CYB_REQUIRE(!need_y || descs[i].n == 0 || descs[i].y, "vector desc %lld: y is NULL", (long long)i);
        CYB_REQUIRE(!need_out || descs[i].n == 0 || descs[i].out, "vector desc %lld: out is NULL", (long long)i);
        hv[(size_t)i] = VecDev{descs[i].x, descs[i].y, descs[i].out, descs[i].n};
    }
    make_items<cyb_vec_desc>(descs, n, items, vec_count);
    if (defer) { // (the caller packs further arrays into the same upload)
        keep->swap(hv);
        return CYB_OK;
    }
    return cyb::upload_packed(ctx, {{hv.data(), sizeof(VecDev) * hv.size(), d_descs}, {items.data(), sizeof(Item) * items.size(), d_items}});
}

// shared body of the three reductions. per_entry: one result per list entry, else one total
static int reduce_common(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double* result_dev, int mode, bool per_entry)
{
    CYB_REQUIRE(ctx && result_dev, "reduction: NULL argument");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "reduction: bad descriptor list");
    const int64_t n_groups = per_entry ? n : 1;
    if (n_groups == 0) return CYB_OK;
    if (n == 1 && descs[0].n > 0) { // ONE vector: no descriptor upload
        CYB_REQUIRE(descs[0].x, "vector desc 0: x is NULL");
        const VecDev d{descs[0].x, descs[0].y, nullptr, descs[0].n};
        const int64_t chunk = chunk_for(d.n), n_items = cdiv64(d.n, chunk);
        void* ws1 = nullptr;
        CYB_TRY(ctx->workspace(sizeof(double) * (size_t)n_items, &ws1));
        hipLaunchKernelGGL(reduce_stage1_one_kernel, dim3((unsigned)n_items), dim3(NT), 0, ctx->stream, d, chunk, static_cast<double*>(ws1), mode);
        hipLaunchKernelGGL(reduce_stage2_one_kernel, dim3(1), dim3(NT), 0, ctx->stream, static_cast<const double*>(ws1), n_items, result_dev, mode);
        CYB_HIP(hipGetLastError());
        return CYB_OK;
    }
    std::vector<Item> items;
    void *d_descs = nullptr, *d_items = nullptr;
    std::vector<VecDev> hv;
    CYB_TRY(upload_vecs(ctx, descs, n, false, false, items, &d_descs, &d_items, true, &hv));
    // segment table for stage 2
    std::vector<int64_t> seg((size_t)n_groups + 1, 0);
    if (per_entry) {
        size_t k = 0;
        for (int64_t i = 0; i < n; ++i) {
            seg[(size_t)i] = (int64_t)k;
            while (k < items.size() && items[k].desc == i) ++k;
        }
        seg[(size_t)n] = (int64_t)items.size();
    } else {
        seg[1] = (int64_t)items.size();
    }
    void* d_seg = nullptr;
    CYB_TRY(cyb::upload_packed(ctx, {{hv.data(), sizeof(VecDev) * hv.size(), &d_descs},
                                {items.data(), sizeof(Item) * items.size(), &d_items},
                                {seg.data(), sizeof(int64_t) * seg.size(), &d_seg}}));
    void* ws = nullptr;
    CYB_TRY(ctx->workspace(sizeof(double) * std::max<size_t>(items.size(), 1), &ws));
    if (!items.empty())
        hipLaunchKernelGGL(reduce_stage1_kernel, dim3((unsigned)items.size()), dim3(NT), 0, ctx->stream,
                           static_cast<const VecDev*>(d_descs), static_cast<const Item*>(d_items), static_cast<double*>(ws), mode);
    hipLaunchKernelGGL(reduce_stage2_kernel, dim3((unsigned)n_groups), dim3(NT), 0, ctx->stream, static_cast<const double*>(ws),
                       static_cast<const int64_t*>(d_seg), result_dev, mode);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

static int elementwise_common(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, int kind, int op, double a, double b,
                              bool need_y)
{
    CYB_REQUIRE(ctx, "elementwise: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "elementwise: bad descriptor list");
    if (n == 0) return CYB_OK;
    if (n == 1 && descs[0].n > 0) { // ONE vector: no descriptor upload
        CYB_REQUIRE(descs[0].x && descs[0].out && (!need_y || descs[0].y), "vector desc 0: NULL operand");
        const VecDev d{descs[0].x, descs[0].y, descs[0].out, descs[0].n};
        const int64_t chunk = chunk_for(d.n);
        hipLaunchKernelGGL(elementwise_one_kernel, dim3((unsigned)cdiv64(d.n, chunk)), dim3(NT), 0, ctx->stream, d, chunk, kind, op, a, b);
        CYB_HIP(hipGetLastError());
        return CYB_OK;
    }
    std::vector<Item> items;
    void *d_descs = nullptr, *d_items = nullptr;
    CYB_TRY(upload_vecs(ctx, descs, n, need_y, true, items, &d_descs, &d_items));
    if (items.empty()) return CYB_OK;
    hipLaunchKernelGGL(elementwise_kernel, dim3((unsigned)items.size()), dim3(NT), 0, ctx->stream,
                       static_cast<const VecDev*>(d_descs), static_cast<const Item*>(d_items), kind, op, a, b);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

} // namespace

extern "C" {

int cyb_copy_strided_batched(cyb_ctx_t ctx, const cyb_copy_desc* descs, int64_t n, int32_t elem_size)
{
    CYB_REQUIRE(ctx, "cyb_copy_strided_batched: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_copy_strided_batched: bad descriptor list");
    CYB_REQUIRE(elem_size == 1 || elem_size == 4 || elem_size == 8 || elem_size == 16,
                "cyb_copy_strided_batched: unsupported elem_size %d", elem_size);
    if (n == 0) return CYB_OK;
    std::vector<CopyDev> hd((size_t)n);
    std::vector<Item> items;
    std::vector<CopyT> ht;    // descriptors that take the tiled transposing path
    std::vector<Item> titems; // their work items (ranges of tiles)
    std::vector<int64_t> pending; // descriptors that take the strided path
    int64_t pending_total = 0;
    for (int64_t i = 0; i < n; ++i) {
        const cyb_copy_desc& d = descs[i];
        CYB_REQUIRE(d.ndim >= 0 && d.ndim <= CYB_MAX_NDIM, "copy desc %lld: ndim %d out of range", (long long)i, d.ndim);
        CYB_REQUIRE(!d.conj || elem_size == 16, "copy desc %lld: conj needs elem_size 16", (long long)i);
        CopyDev& c = hd[(size_t)i];
        c.dst = d.dst;
        c.src = d.src;
        c.conj = d.conj;
        int64_t tot = 1;
        // drop singleton axes and merge axes that are contiguous in BOTH operands
        int nd = 0;
        for (int k = 0; k < d.ndim; ++k) {
            CYB_REQUIRE(d.shape[k] >= 0, "copy desc %lld: negative extent", (long long)i);
            tot *= d.shape[k];
            if (d.shape[k] == 1) continue;
            if (nd > 0 && c.ds[nd - 1] == d.dst_strides[k] * d.shape[k] && c.ss[nd - 1] == d.src_strides[k] * d.shape[k]) {
                c.shape[nd - 1] *= d.shape[k];
                c.ds[nd - 1] = d.dst_strides[k];
                c.ss[nd - 1] = d.src_strides[k];
            } else {
                c.shape[nd] = d.shape[k];
                c.ds[nd] = d.dst_strides[k];
                c.ss[nd] = d.src_strides[k];
                ++nd;
            }
        }
        for (int k = nd; k < CYB_MAX_NDIM; ++k) {
            c.shape[k] = 1;
            c.ds[k] = c.ss[k] = 0;
        }
        c.ndim = nd;
        c.total = tot;
        CYB_REQUIRE(tot == 0 || (d.dst && d.src), "copy desc %lld: NULL pointer", (long long)i);
        // transposing copy?  (unit-stride axes of source and destination differ and are both long enough)
        int aS = -1, aD = -1;
        for (int k = 0; k < nd; ++k) {
            if (c.ss[k] == 1 && aS < 0) aS = k;
            if (c.ds[k] == 1 && aD < 0) aD = k;
        }
        static const bool no_tiled = getenv("CYB_COPY_NOTILED") != nullptr;
        // a short unit-stride axis may be flattened with the axis that is next-contiguous on the same side
        int pD = -1, pS = -1; // partner axes (outer halves of the composites)
        if (aS >= 0 && aD >= 0 && aS != aD) {
            if (c.shape[aD] < 16)
                for (int k = 0; k < nd; ++k)
                    if (k != aD && k != aS && c.ds[k] == c.shape[aD]) pD = k;
            if (c.shape[aS] < 16)
                for (int k = 0; k < nd; ++k)
                    if (k != aS && k != aD && k != pD && c.ss[k] == c.shape[aS]) pS = k;
        }
        const int64_t extS = aS >= 0 ? c.shape[aS] * (pS >= 0 ? c.shape[pS] : 1) : 0;
        const int64_t extD = aD >= 0 ? c.shape[aD] * (pD >= 0 ? c.shape[pD] : 1) : 0;
        if (!no_tiled && tot > 0 && aS >= 0 && aD >= 0 && aS != aD && extS >= 16 && extD >= 16) {
            CopyT t;
            memset(&t, 0, sizeof(t));
            t.dst = d.dst;
            t.src = d.src;
            t.conj = d.conj;
            t.nS = extS;
            t.nD = extD;
            if (pD >= 0) {
                t.nD2 = c.shape[aD];
                t.ssD = c.ss[pD];
                t.ssD2 = c.ss[aD];
            } else {
                t.nD2 = 1;
                t.ssD = c.ss[aD];
                t.ssD2 = 0;
            }
            if (pS >= 0) {
                t.nS2 = c.shape[aS];
                t.dsS = c.ds[pS];
                t.dsS2 = c.ds[aS];
            } else {
                t.nS2 = 1;
                t.dsS = c.ds[aS];
                t.dsS2 = 0;
            }
            const int64_t tsz = (elem_size == 8 && !d.conj) ? 64 : 32;
            t.tilesS = (t.nS + tsz - 1) / tsz;
            t.tilesD = (t.nD + tsz - 1) / tsz;
            int64_t outer = 1;
            for (int k = 0; k < nd; ++k) {
                if (k == aS || k == aD || k == pS || k == pD) continue;
                t.oshape[t.n_outer] = c.shape[k];
                t.ods[t.n_outer] = c.ds[k];
                t.oss[t.n_outer] = c.ss[k];
                ++t.n_outer;
                outer *= c.shape[k];
            }
            const int64_t ntile = outer * t.tilesS * t.tilesD;
            const int64_t kTilesPerItem = tsz == 64 ? 4 : 16;
            for (int64_t s0 = 0; s0 < ntile; s0 += kTilesPerItem)
                titems.push_back(Item{(int32_t)ht.size(), 0, s0, std::min(kTilesPerItem, ntile - s0)});
            ht.push_back(t);
            continue;
        }
        pending.push_back(i);
        pending_total += tot;
    }
    const int64_t chunk = chunk_for(pending_total);
    for (int64_t i : pending) {
        const int64_t tot = hd[(size_t)i].total;
        for (int64_t s = 0; s < tot; s += chunk) items.push_back(Item{(int32_t)i, 0, s, std::min(chunk, tot - s)});
    }
    if (!titems.empty()) {
        void *d_t = nullptr, *d_ti = nullptr;
        CYB_TRY(cyb::upload_packed(ctx, {{ht.data(), sizeof(CopyT) * ht.size(), &d_t}, {titems.data(), sizeof(Item) * titems.size(), &d_ti}}));
        const dim3 tgrid((unsigned)titems.size()), tblock(NT);
        const CopyT* dt = static_cast<const CopyT*>(d_t);
        const Item* dti = static_cast<const Item*>(d_ti);
        switch (elem_size) {
        case 1: hipLaunchKernelGGL(copy_transpose_kernel<uint8_t>, tgrid, tblock, 0, ctx->stream, dt, dti); break;
        case 4: hipLaunchKernelGGL(copy_transpose_kernel<uint32_t>, tgrid, tblock, 0, ctx->stream, dt, dti); break;
        case 8: hipLaunchKernelGGL(copy_transpose64_kernel, tgrid, tblock, 0, ctx->stream, dt, dti); break;
        default: hipLaunchKernelGGL(copy_transpose_kernel<u128>, tgrid, tblock, 0, ctx->stream, dt, dti); break;
        }
        CYB_HIP(hipGetLastError());
    }
    if (items.empty()) return CYB_OK;
    void *d_descs = nullptr, *d_items = nullptr;
    CYB_TRY(cyb::upload_packed(ctx, {{hd.data(), sizeof(CopyDev) * hd.size(), &d_descs}, {items.data(), sizeof(Item) * items.size(), &d_items}}));
    const dim3 grid((unsigned)items.size()), block(NT);
    const CopyDev* dd = static_cast<const CopyDev*>(d_descs);
    const Item* di = static_cast<const Item*>(d_items);
    switch (elem_size) {
    case 1: hipLaunchKernelGGL(copy_strided_kernel<uint8_t>, grid, block, 0, ctx->stream, dd, di); break;
    case 4: hipLaunchKernelGGL(copy_strided_kernel<uint32_t>, grid, block, 0, ctx->stream, dd, di); break;
    case 8: hipLaunchKernelGGL(copy_strided_kernel<uint64_t>, grid, block, 0, ctx->stream, dd, di); break;
    default: hipLaunchKernelGGL(copy_strided_kernel<u128>, grid, block, 0, ctx->stream, dd, di); break;
    }
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_dot_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double* result_dev)
{
    return reduce_common(ctx, descs, n, result_dev, 0, false);
}
int cyb_dot_each_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double* result_dev)
{
    return reduce_common(ctx, descs, n, result_dev, 0, true);
}
int cyb_maxabs_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double* result_dev)
{
    return reduce_common(ctx, descs, n, result_dev, 1, false);
}
int cyb_axpby_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double a, double b)
{
    return elementwise_common(ctx, descs, n, 0, 0, a, b, false);
}
int cyb_binary_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, int32_t op)
{
    CYB_REQUIRE(op >= 0 && op <= 4, "cyb_binary_batched_f64: unknown op %d", op);
    return elementwise_common(ctx, descs, n, 1, op, 0, 0, true);
}
int cyb_unary_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, int32_t op)
{
    CYB_REQUIRE(op >= 0 && op <= 8, "cyb_unary_batched_f64: unknown op %d", op);
    return elementwise_common(ctx, descs, n, 2, op, 0, 0, false);
}

int cyb_unary_param_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, int32_t op, double param)
{
    CYB_REQUIRE(op >= 0 && op <= 3, "cyb_unary_param_batched_f64: unknown op %d", op);
    return elementwise_common(ctx, descs, n, 3, op, param, 0, false);
}

int cyb_compare_f64(cyb_ctx_t ctx, const double* x, const double* y, double scalar, uint8_t* out, int64_t n, int32_t op)
{
    CYB_REQUIRE(ctx && n >= 0 && (n == 0 || (x && out)), "cyb_compare_f64: bad argument");
    CYB_REQUIRE(op >= 0 && op <= 5, "cyb_compare_f64: unknown op %d", op);
    if (n == 0) return CYB_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n, NT * 4), 4096);
    hipLaunchKernelGGL(compare_kernel, dim3(grid), dim3(NT), 0, ctx->stream, x, y, scalar, out, n, op);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_convert_u8_f64(cyb_ctx_t ctx, const uint8_t* x, double* out, int64_t n)
{
    CYB_REQUIRE(ctx && n >= 0 && (n == 0 || (x && out)), "cyb_convert_u8_f64: bad argument");
    if (n == 0) return CYB_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n, NT * 4), 4096);
    hipLaunchKernelGGL(convert_u8_f64_kernel, dim3(grid), dim3(NT), 0, ctx->stream, x, out, n);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_count_nonzero_u8(cyb_ctx_t ctx, const uint8_t* x, int64_t n, uint64_t* result_dev)
{
    CYB_REQUIRE(ctx && result_dev && n >= 0 && (n == 0 || x), "cyb_count_nonzero_u8: bad argument");
    CYB_HIP(hipMemsetAsync(result_dev, 0, sizeof(uint64_t), ctx->stream));
    if (n == 0) return CYB_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n, NT * 16), 2048);
    hipLaunchKernelGGL(count_nonzero_kernel, dim3(grid), dim3(NT), 0, ctx->stream, x, n,
                       reinterpret_cast<unsigned long long*>(result_dev));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_extremum_f64(cyb_ctx_t ctx, const double* x, int64_t n, int32_t mode, double* result_dev)
{
    CYB_REQUIRE(ctx && result_dev && n >= 1 && x, "cyb_extremum_f64: needs a non-empty vector");
    CYB_REQUIRE(mode >= 0 && mode <= 2, "cyb_extremum_f64: unknown mode %d", mode);
    const int64_t grid = std::min<int64_t>(cdiv64(n, NT * 8), 1024);
    void* part_v = nullptr;
    CYB_TRY(ctx->workspace(sizeof(Ext) * (size_t)grid, &part_v, 2));
    Ext* part = static_cast<Ext*>(part_v);
    hipLaunchKernelGGL(extremum_kernel, dim3((unsigned)grid), dim3(NT), 0, ctx->stream, x, n, mode, part, (const Ext*)nullptr, (int64_t)0);
    // result_dev[0] = key of the extremum, result_dev[1] holds the flat index as an int64 bit pattern
    static_assert(sizeof(Ext) == 2 * sizeof(double), "Ext is written into two doubles");
    hipLaunchKernelGGL(extremum_kernel, dim3(1), dim3(NT), 0, ctx->stream, x, n, mode, reinterpret_cast<Ext*>(result_dev),
                       (const Ext*)part, grid);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_random_uniform_f64(cyb_ctx_t ctx, double* out, int64_t n, uint64_t seed, double lo, double hi)
{
    CYB_REQUIRE(ctx && (n == 0 || out) && n >= 0, "cyb_random_uniform_f64: bad argument");
    if (n == 0) return CYB_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64((n + 1) / 2, NT), 2048);
    hipLaunchKernelGGL(random_uniform_kernel, dim3(grid), dim3(NT), 0, ctx->stream, out, n, seed, lo, hi);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

extern "C++" {
template <class TermIn, class TermDev, class Fill>
static int lincomb_common(cyb_ctx_t ctx, const cyb_lincomb_desc* descs, int64_t n, const TermIn* terms, int64_t n_terms, Fill fill,
                          void (*kernel)(const LinDev*, const TermDev*, const Item*))
{
    CYB_REQUIRE(ctx, "cyb_lincomb_strided_batched: ctx is NULL");
    CYB_REQUIRE(n >= 0 && n_terms >= 0 && (n == 0 || descs) && (n_terms == 0 || terms), "cyb_lincomb_strided_batched: bad lists");
    if (n == 0) return CYB_OK;
    std::vector<LinDev> hd((size_t)n);
    std::vector<TermDev> ht((size_t)n_terms);
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) {
        const cyb_lincomb_desc& d = descs[i];
        CYB_REQUIRE(d.ndim >= 0 && d.ndim <= CYB_MAX_NDIM, "lincomb desc %lld: ndim %d out of range", (long long)i, d.ndim);
        CYB_REQUIRE(d.term_begin >= 0 && d.term_begin <= d.term_end && d.term_end <= n_terms,
                    "lincomb desc %lld: bad term range [%d,%d)", (long long)i, d.term_begin, d.term_end);
        LinDev& c = hd[(size_t)i];
        c.dst = d.dst;
        c.ndim = d.ndim;
        c.accumulate = d.accumulate;
        c.term_begin = d.term_begin;
        c.term_end = d.term_end;
        int64_t tot = 1;
        for (int k = 0; k < CYB_MAX_NDIM; ++k) {
            c.shape[k] = k < d.ndim ? d.shape[k] : 1;
            c.ds[k] = k < d.ndim ? d.dst_strides[k] : 0;
            CYB_REQUIRE(c.shape[k] >= 0, "lincomb desc %lld: negative extent", (long long)i);
            tot *= c.shape[k];
        }
        c.total = tot;
        CYB_REQUIRE(tot == 0 || d.dst, "lincomb desc %lld: dst is NULL", (long long)i);
        for (int32_t t = d.term_begin; t < d.term_end; ++t)
            CYB_REQUIRE(tot == 0 || terms[t].src, "lincomb term %d: src is NULL", t);
        total += tot;
    }
    for (int64_t t = 0; t < n_terms; ++t) fill(ht[(size_t)t], terms[t]);
    std::vector<Item> items;
    const int64_t chunk = chunk_for(total);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t s0 = 0; s0 < hd[(size_t)i].total; s0 += chunk)
            items.push_back(Item{(int32_t)i, 0, s0, std::min(chunk, hd[(size_t)i].total - s0)});
    if (items.empty()) return CYB_OK;
    void *d_descs = nullptr, *d_terms = nullptr, *d_items = nullptr;
    CYB_TRY(cyb::upload_packed(ctx, {{hd.data(), sizeof(LinDev) * hd.size(), &d_descs},
                                {ht.empty() ? (const void*)hd.data() : (const void*)ht.data(), ht.empty() ? 8 : sizeof(TermDev) * ht.size(), &d_terms},
                                {items.data(), sizeof(Item) * items.size(), &d_items}}));
    hipLaunchKernelGGL(kernel, dim3((unsigned)items.size()), dim3(NT), 0, ctx->stream,
                       static_cast<const LinDev*>(d_descs), static_cast<const TermDev*>(d_terms), static_cast<const Item*>(d_items));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

} // extern "C++"

int cyb_lincomb_strided_batched_f64(cyb_ctx_t ctx, const cyb_lincomb_desc* descs, int64_t n, const cyb_lincomb_term* terms,
                                    int64_t n_terms)
{
    return lincomb_common<cyb_lincomb_term, LinTerm>(
        ctx, descs, n, terms, n_terms,
        [](LinTerm& o, const cyb_lincomb_term& t) {
            o.src = t.src;
            o.coeff = t.coeff;
            for (int k = 0; k < CYB_MAX_NDIM; ++k) o.ss[k] = t.src_strides[k];
        },
        lincomb_strided_kernel);
}

int cyb_lincomb_strided_batched_c128(cyb_ctx_t ctx, const cyb_lincomb_desc* descs, int64_t n, const cyb_lincomb_term_c128* terms,
                                     int64_t n_terms)
{
    return lincomb_common<cyb_lincomb_term_c128, LinTermC>(
        ctx, descs, n, terms, n_terms,
        [](LinTermC& o, const cyb_lincomb_term_c128& t) {
            o.src = t.src;
            o.cr = t.coeff_re;
            o.ci = t.coeff_im;
            o.src_real = t.src_real;
            o.pad = 0;
            for (int k = 0; k < CYB_MAX_NDIM; ++k) o.ss[k] = t.src_strides[k];
        },
        lincomb_strided_c128_kernel);
}

int cyb_scale_axis_batched_f64(cyb_ctx_t ctx, const cyb_scale_axis_desc* descs, int64_t n)
{
    CYB_REQUIRE(ctx, "cyb_scale_axis_batched_f64: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_scale_axis_batched_f64: bad descriptor list");
    if (n == 0) return CYB_OK;
    std::vector<ScaleDev> hd((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const auto& d = descs[i];
        CYB_REQUIRE(d.outer >= 0 && d.axis >= 0 && d.inner >= 0, "scale_axis desc %lld: negative extent", (long long)i);
        CYB_REQUIRE(d.outer * d.axis * d.inner == 0 || (d.x && d.f && d.out), "scale_axis desc %lld: NULL pointer", (long long)i);
        hd[(size_t)i] = ScaleDev{d.x, d.f, d.out, d.outer, d.axis, d.inner};
    }
    std::vector<Item> items;
    make_items<cyb_scale_axis_desc>(descs, n, items, scale_count);
    if (items.empty()) return CYB_OK;
    void *d_descs = nullptr, *d_items = nullptr;
    CYB_TRY(cyb::upload_packed(ctx, {{hd.data(), sizeof(ScaleDev) * hd.size(), &d_descs}, {items.data(), sizeof(Item) * items.size(), &d_items}}));
    hipLaunchKernelGGL(scale_axis_kernel, dim3((unsigned)items.size()), dim3(NT), 0, ctx->stream,
                       static_cast<const ScaleDev*>(d_descs), static_cast<const Item*>(d_items));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

static int mask_common(cyb_ctx_t ctx, const cyb_mask_desc* descs, int64_t n, int scatter)
{
    CYB_REQUIRE(ctx, "mask op: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "mask op: bad descriptor list");
    if (n == 0) return CYB_OK;
    std::vector<MaskDev> hd((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const auto& d = descs[i];
        CYB_REQUIRE(d.outer >= 0 && d.axis >= 0 && d.inner >= 0 && d.n_keep >= 0 && d.n_keep <= d.axis,
                    "mask desc %lld: bad extents", (long long)i);
        CYB_REQUIRE(d.outer * d.n_keep * d.inner == 0 || (d.x && d.out && d.idx), "mask desc %lld: NULL pointer", (long long)i);
        hd[(size_t)i] = MaskDev{d.x, d.out, d.idx, d.outer, d.axis, d.inner, d.n_keep};
        if (scatter && d.outer * d.axis * d.inner > 0)
            CYB_HIP(hipMemsetAsync(d.out, 0, sizeof(double) * (size_t)(d.outer * d.axis * d.inner), ctx->stream));
    }
    std::vector<Item> items;
    make_items<cyb_mask_desc>(descs, n, items, mask_count);
    if (items.empty()) return CYB_OK;
    void *d_descs = nullptr, *d_items = nullptr;
    CYB_TRY(cyb::upload_packed(ctx, {{hd.data(), sizeof(MaskDev) * hd.size(), &d_descs}, {items.data(), sizeof(Item) * items.size(), &d_items}}));
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)items.size()), dim3(NT), 0, ctx->stream,
                       static_cast<const MaskDev*>(d_descs), static_cast<const Item*>(d_items), scatter);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_mask_gather_batched_f64(cyb_ctx_t ctx, const cyb_mask_desc* descs, int64_t n) { return mask_common(ctx, descs, n, 0); }
int cyb_mask_scatter_batched_f64(cyb_ctx_t ctx, const cyb_mask_desc* descs, int64_t n) { return mask_common(ctx, descs, n, 1); }

int cyb_fill_f64(cyb_ctx_t ctx, double* out, int64_t n, double value)
{
    CYB_REQUIRE(ctx && (n == 0 || out) && n >= 0, "cyb_fill_f64: bad argument");
    if (n == 0) return CYB_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n, NT), 2048);
    hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(NT), 0, ctx->stream, out, n, value);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_eye_f64(cyb_ctx_t ctx, double* out, int64_t n)
{
    CYB_REQUIRE(ctx && (n == 0 || out) && n >= 0, "cyb_eye_f64: bad argument");
    if (n == 0) return CYB_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n * n, NT), 2048);
    hipLaunchKernelGGL(eye_kernel, dim3(grid), dim3(NT), 0, ctx->stream, out, n);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_random_normal_f64(cyb_ctx_t ctx, double* out, int64_t n, uint64_t seed, double sigma)
{
    CYB_REQUIRE(ctx && (n == 0 || out) && n >= 0, "cyb_random_normal_f64: bad argument");
    if (n == 0) return CYB_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64((n + 1) / 2, NT), 2048);
    hipLaunchKernelGGL(random_normal_kernel, dim3(grid), dim3(NT), 0, ctx->stream, out, n, seed, sigma);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_complex_expand_batched_f64(cyb_ctx_t ctx, const cyb_cexpand_desc* descs, int64_t n)
{
    CYB_REQUIRE(ctx, "cyb_complex_expand_batched_f64: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_complex_expand_batched_f64: bad descriptor list");
    if (n == 0) return CYB_OK;
    std::vector<CExpandDev> hd((size_t)n);
    std::vector<Item> items;
    for (int64_t i = 0; i < n; ++i) {
        const cyb_cexpand_desc& d = descs[i];
        CYB_REQUIRE(d.K >= 0 && d.N >= 0, "complex expand desc %lld: negative extent", (long long)i);
        const int64_t tot = d.K * d.N;
        CYB_REQUIRE(tot == 0 || (d.src && d.dst), "complex expand desc %lld: NULL pointer", (long long)i);
        hd[(size_t)i] = CExpandDev{d.src, d.rs, d.cs, d.K, d.N, d.dst};
        for (int64_t s0 = 0; s0 < tot; s0 += CHUNK) items.push_back(Item{(int32_t)i, 0, s0, std::min(CHUNK, tot - s0)});
    }
    if (items.empty()) return CYB_OK;
    void *d_descs = nullptr, *d_items = nullptr;
    CYB_TRY(cyb::upload_packed(ctx, {{hd.data(), sizeof(CExpandDev) * hd.size(), &d_descs}, {items.data(), sizeof(Item) * items.size(), &d_items}}));
    hipLaunchKernelGGL(complex_expand_kernel, dim3((unsigned)items.size()), dim3(NT), 0, ctx->stream,
                       static_cast<const CExpandDev*>(d_descs), static_cast<const Item*>(d_items));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_elementwise_batched_c128(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, int32_t op)
{
    CYB_REQUIRE(ctx, "cyb_elementwise_batched_c128: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_elementwise_batched_c128: bad descriptor list");
    CYB_REQUIRE(op >= 0 && op <= 6, "cyb_elementwise_batched_c128: unknown op %d", op);
    std::vector<Item> items;
    void *d_descs = nullptr, *d_items = nullptr;
    CYB_TRY(upload_vecs(ctx, descs, n, op >= 5, true, items, &d_descs, &d_items));
    if (items.empty()) return CYB_OK;
    hipLaunchKernelGGL(celementwise_kernel, dim3((unsigned)items.size()), dim3(NT), 0, ctx->stream,
                       static_cast<const VecDev*>(d_descs), static_cast<const Item*>(d_items), op);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int cyb_axpby_batched_c128(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double a_re, double a_im, double b_re,
                           double b_im)
{
    CYB_REQUIRE(ctx, "cyb_axpby_batched_c128: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_axpby_batched_c128: bad descriptor list");
    std::vector<Item> items;
    void *d_descs = nullptr, *d_items = nullptr;
    CYB_TRY(upload_vecs(ctx, descs, n, false, true, items, &d_descs, &d_items));
    if (items.empty()) return CYB_OK;
    hipLaunchKernelGGL(axpby_c128_kernel, dim3((unsigned)items.size()), dim3(NT), 0, ctx->stream,
                       static_cast<const VecDev*>(d_descs), static_cast<const Item*>(d_items), a_re, a_im, b_re, b_im);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

} // extern "C"
