// complex128 QR by Householder reflections (gfx950) -- the factorisation behind `cyb_qr_batched_c128`.
//
// scipy.linalg.qr(a, mode='economic' | 'full') on complex128 blocks (NumpyBlockBackend::matrix_qr, numpy.cpp:1236-1245:
// LAPACK zgeqrf + zungqr; matrix_lq is the same on the transposed view, block_backend.cpp:1033-1040).  Householder QR is
// backward stable for EVERY block -- rank deficient, exact copies of columns, zero columns, graded -- which the
// Gram-Schmidt kernels of rounds 1 / 2 were not (a numerically dependent column in the middle of a low-rank block came
// back with |Q^H Q - 1| = 0.9).  Large well-conditioned blocks still take the faster embedded route on the MFMA block
// engine first (block_backend.py `_complex_qr_embedded`); this file is the route for everything else and the
// per-block fallback of that one.
//
// Reflector of column j as LAPACK zlarfg builds it: alpha = x[0], beta = -sign(Re alpha) * ||x||  (real), tau = (beta -
// alpha) / beta (complex), v = x / (alpha - beta) with v[0] = 1, H = I - tau v v^H; tau = 0 (H = I) only for a column
// whose tail is exactly zero and whose pivot is real.  The trailing columns get H^H (zgeqr2), Q = H_0 H_1 ... H_{k-1}
// (zungqr).  On write-out row j of R and column j of Q are multiplied by sign(beta_j), so diag(R) >= 0 -- the convention
// of the embedded route, i.e. the same factorisation whichever route a block took.
//
// Layout: column-major working copy W (m x n); column j keeps its ORIGINAL tail below the diagonal for the whole
// factorisation (v is x * scale_j, recomputed on the fly), so a step reads column j and writes only columns > j.
//   * blocks of at most kSingleElems elements: ONE launch, one workgroup per block loops over the columns;
//   * larger blocks: one launch per column step for the whole list (grid.y = block): every workgroup derives the
//     reflector of column j redundantly (same code, same order: bit-identical) and updates its share of the trailing
//     columns, one wave per column; then one launch per step of the backward accumulation of Q.
#include "common.h"

#include <algorithm>
#include <vector>

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
constexpr int NW = NT / 64;
constexpr int64_t kSingleElems = 96 * 96;

struct CBlk {
    const d2* A;
    d2 *Q, *R;
    int64_t lda, ldq, ldr;
    d2* W;        // m x n column-major working copy
    d2* Qc;       // m x kq column-major Q
    d2* tau;      // k
    d2* scale;    // k: v[1:] = x[1:] * scale
    double* beta; // k
    int32_t m, n, k, kq;
};

__device__ __forceinline__ d2 cmul(d2 a, d2 b) { return d2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// conj(a) * b
__device__ __forceinline__ d2 cmulc(d2 a, d2 b) { return d2{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x}; }

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

struct Refl {
    d2 tau, scale;
    double beta;
};

// Reflector of x[0 .. L) (zlarfg), computed by the whole workgroup; every thread returns the same values.  `red` holds
// NW doubles.  Deterministic: a fixed partition of the tail over the threads and a fixed summation order.
__device__ Refl make_reflector(const d2* __restrict__ x, int L, double* red)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double mx = 0.0;
    for (int i = 1 + tid; i < L; i += NT) {
        const d2 v = x[i];
        mx = fmax(mx, fmax(fabs(v.x), fabs(v.y)));
    }
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = 0.0;
#pragma unroll
    for (int q = 0; q < NW; ++q) mx = fmax(mx, red[q]);
    double ss = 0.0;
    const bool tail = mx > 1e-290; // (a denormal tail is zero: 1 / mx would overflow)
    if (tail) {
        const double inv = 1.0 / mx;
        for (int i = 1 + tid; i < L; i += NT) {
            const d2 v = x[i] * inv;
            ss += v.x * v.x + v.y * v.y;
        }
    }
    ss = wave_sum(ss);
    __syncthreads();
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < NW; ++q) tot += red[q];
    const double xnorm = tail ? mx * sqrt(tot) : 0.0;
    const d2 alpha = x[0];
    Refl r;
    if (xnorm == 0.0 && alpha.y == 0.0) {
        r.tau = d2{0.0, 0.0};
        r.scale = d2{0.0, 0.0};
        r.beta = alpha.x;
        return r;
    }
    const double m3 = fmax(fmax(fabs(alpha.x), fabs(alpha.y)), xnorm);
    if (m3 < 1e-290) { // a column at the denormal level (the residue of ~19 eliminations of an exactly dependent column) is zero:
        r.tau = d2{0.0, 0.0}; // quotients of denormals carry no accuracy and tau would not be that of a unitary H
        r.scale = d2{0.0, 0.0};
        r.beta = 0.0;
        return r;
    }
    const double a0 = alpha.x / m3, a1 = alpha.y / m3, a2 = xnorm / m3;
    const double beta = -copysign(m3 * sqrt(a0 * a0 + a1 * a1 + a2 * a2), alpha.x);
    r.beta = beta;
    r.tau = d2{(beta - alpha.x) / beta, -alpha.y / beta};
    // scale = 1 / (alpha - beta), |alpha.x - beta| >= |beta|: no cancellation; Smith-style against overflow of |z|^2
    const double zr = alpha.x - beta, zi = alpha.y;
    const double zm = fmax(fabs(zr), fabs(zi));
    const double pr = zr / zm, pi = zi / zm;
    const double den = (pr * pr + pi * pi) * zm;
    r.scale = tail ? d2{pr / den, -pi / den} : d2{0.0, 0.0};
    return r;
}

// a[0 .. L) -= coef * v * (v^H a) for the columns c = c0, c0 + cs, ... < c1 of the column-major array `base` (column
// stride ld, rows starting at the reflector's pivot row): one wave per column.  v[0] = 1, v[i] = x[i] * scale.
__device__ void apply_reflector(d2* __restrict__ base, int64_t ld, const d2* __restrict__ x, d2 scale, d2 coef, int L, int c0, int c1, int cs)
{
    const int lane = threadIdx.x & 63;
    for (int c = c0; c < c1; c += cs) {
        d2* a = base + (int64_t)c * ld;
        d2 dot = (lane == 0) ? a[0] : d2{0.0, 0.0};
        for (int i = 1 + lane; i < L; i += 64) dot += cmulc(cmul(x[i], scale), a[i]);
        dot.x = wave_sum(dot.x);
        dot.y = wave_sum(dot.y);
        const d2 w = cmul(coef, dot);
        if (lane == 0) a[0] -= w;
        for (int i = 1 + lane; i < L; i += 64) a[i] -= cmul(cmul(x[i], scale), w);
    }
}

__device__ void load_block(const CBlk& d, int64_t start, int64_t stride)
{
    const int64_t m = d.m, tot = m * d.n;
    for (int64_t e = start; e < tot; e += stride) {
        const int64_t c = e / m, i = e - c * m;
        d.W[e] = d.A[i * d.lda + c];
    }
}

__device__ void init_q(const CBlk& d, int64_t start, int64_t stride)
{
    const int64_t m = d.m, tot = m * d.kq;
    for (int64_t e = start; e < tot; e += stride) {
        const int64_t c = e / m, i = e - c * m;
        d.Qc[e] = d2{c == i ? 1.0 : 0.0, 0.0};
    }
}

// R (kq x n, upper triangular / trapezoidal, zero rows beyond k) and Q (m x kq), rows of R / columns of Q signed so that
// diag(R) >= 0
__device__ void write_block(const CBlk& d, int64_t start, int64_t stride)
{
    const int64_t n = d.n, kq = d.kq, m = d.m;
    for (int64_t e = start; e < kq * n; e += stride) {
        const int64_t i = e / n, c = e - i * n;
        d2 v = d2{0.0, 0.0};
        if (i < d.k && i <= c) {
            const double s = d.beta[i] < 0.0 ? -1.0 : 1.0;
            v = (i == c) ? d2{fabs(d.beta[i]), 0.0} : d.W[c * m + i] * s;
        }
        d.R[i * d.ldr + c] = v;
    }
    for (int64_t e = start; e < m * kq; e += stride) {
        const int64_t i = e / kq, c = e - i * kq;
        const double s = (c < d.k && d.beta[c] < 0.0) ? -1.0 : 1.0;
        d.Q[i * d.ldq + c] = d.Qc[c * m + i] * s;
    }
}

// ---- small blocks: the whole factorisation in one workgroup
__global__ void __launch_bounds__(NT) cqh_single_kernel(const CBlk* __restrict__ blks)
{
    __shared__ double red[NW];
    const CBlk d = blks[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int m = d.m;
    load_block(d, tid, NT);
    __syncthreads();
    for (int j = 0; j < d.k; ++j) {
        const d2* x = d.W + (int64_t)j * m + j;
        const Refl r = make_reflector(x, m - j, red);
        if (tid == 0) {
            d.tau[j] = r.tau;
            d.scale[j] = r.scale;
            d.beta[j] = r.beta;
        }
        if (r.tau.x != 0.0 || r.tau.y != 0.0)
            apply_reflector(d.W + j, m, x, r.scale, d2{r.tau.x, -r.tau.y}, m - j, j + 1 + wave, d.n, NW);
        __syncthreads();
    }
    init_q(d, tid, NT);
    __syncthreads();
    for (int j = d.k - 1; j >= 0; --j) {
        const d2 tau = d.tau[j];
        if (tau.x != 0.0 || tau.y != 0.0)
            apply_reflector(d.Qc + j, m, d.W + (int64_t)j * m + j, d.scale[j], tau, m - j, j + wave, d.kq, NW);
        __syncthreads();
    }
    write_block(d, tid, NT);
}

// ---- large blocks: one launch per step
__global__ void __launch_bounds__(NT) cqh_load_kernel(const CBlk* __restrict__ blks)
{
    const CBlk d = blks[blockIdx.y];
    load_block(d, (int64_t)blockIdx.x * NT + threadIdx.x, (int64_t)gridDim.x * NT);
    init_q(d, (int64_t)blockIdx.x * NT + threadIdx.x, (int64_t)gridDim.x * NT);
}

__global__ void __launch_bounds__(NT) cqh_step_kernel(const CBlk* __restrict__ blks, int j)
{
    __shared__ double red[NW];
    const CBlk d = blks[blockIdx.y];
    if (j >= d.k) return;
    const int trailing = d.n - j - 1;
    // (workgroup 0 always runs: it records the reflector)
    if (blockIdx.x != 0 && (int)blockIdx.x * NW >= trailing) return;
    const int m = d.m;
    const d2* x = d.W + (int64_t)j * m + j;
    const Refl r = make_reflector(x, m - j, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.tau[j] = r.tau;
        d.scale[j] = r.scale;
        d.beta[j] = r.beta;
    }
    if (r.tau.x != 0.0 || r.tau.y != 0.0)
        apply_reflector(d.W + j, m, x, r.scale, d2{r.tau.x, -r.tau.y}, m - j, j + 1 + (int)blockIdx.x * NW + (int)(threadIdx.x >> 6), d.n,
                        (int)gridDim.x * NW);
}

__global__ void __launch_bounds__(NT) cqh_qstep_kernel(const CBlk* __restrict__ blks, int j)
{
    const CBlk d = blks[blockIdx.y];
    if (j >= d.k) return;
    const d2 tau = d.tau[j];
    if (tau.x == 0.0 && tau.y == 0.0) return;
    const int m = d.m;
    apply_reflector(d.Qc + j, m, d.W + (int64_t)j * m + j, d.scale[j], tau, m - j, j + (int)blockIdx.x * NW + (int)(threadIdx.x >> 6), d.kq,
                    (int)gridDim.x * NW);
}

__global__ void __launch_bounds__(NT) cqh_write_kernel(const CBlk* __restrict__ blks)
{
    const CBlk d = blks[blockIdx.y];
    write_block(d, (int64_t)blockIdx.x * NT + threadIdx.x, (int64_t)gridDim.x * NT);
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

} // namespace

extern "C" int cyb_qr_batched_c128(cyb_ctx_t ctx, const cyb_qr_desc* descs, int64_t n)
{
    CYB_REQUIRE(ctx, "cyb_qr_batched_c128: ctx is NULL");
    CYB_REQUIRE(n >= 0 && (n == 0 || descs), "cyb_qr_batched_c128: bad descriptor list");
    std::vector<CBlk> small, large;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes);
        return o;
    };
    for (int64_t i = 0; i < n; ++i) {
        const cyb_qr_desc& s = descs[i];
        CYB_REQUIRE(s.m >= 0 && s.n >= 0, "qr block %lld: negative extent", (long long)i);
        const int64_t k = std::min(s.m, s.n), kq = s.full ? s.m : k;
        if (s.m == 0 || kq == 0) continue;
        CYB_REQUIRE(s.A || s.n == 0, "qr block %lld: A is NULL", (long long)i);
        CYB_REQUIRE(s.Q && (s.R || s.n == 0), "qr block %lld: NULL output", (long long)i);
        CYB_REQUIRE(s.lda >= s.n && s.ldq >= kq && s.ldr >= s.n, "qr block %lld: leading dimension too small", (long long)i);
        CYB_REQUIRE(s.m < (1 << 30) && s.n < (1 << 30), "qr block %lld: extent too large", (long long)i);
        CBlk b;
        b.A = reinterpret_cast<const d2*>(s.A);
        b.Q = reinterpret_cast<d2*>(s.Q);
        b.R = reinterpret_cast<d2*>(s.R);
        b.lda = s.lda;
        b.ldq = s.ldq;
        b.ldr = s.ldr;
        b.m = (int32_t)s.m;
        b.n = (int32_t)s.n;
        b.k = (int32_t)k;
        b.kq = (int32_t)kq;
        // workspace offsets first (pointers once the workspace is known)
        b.W = reinterpret_cast<d2*>(take(sizeof(d2) * (size_t)s.m * (size_t)s.n));
        b.Qc = reinterpret_cast<d2*>(take(sizeof(d2) * (size_t)s.m * (size_t)kq));
        b.tau = reinterpret_cast<d2*>(take(sizeof(d2) * (size_t)std::max<int64_t>(k, 1)));
        b.scale = reinterpret_cast<d2*>(take(sizeof(d2) * (size_t)std::max<int64_t>(k, 1)));
        b.beta = reinterpret_cast<double*>(take(sizeof(double) * (size_t)std::max<int64_t>(k, 1)));
        (s.m * std::max(s.n, kq) <= kSingleElems ? small : large).push_back(b);
    }
    if (small.empty() && large.empty()) return CYB_OK;
    void* ws = nullptr;
    CYB_TRY(ctx->workspace(off, &ws, 0));
    char* base = static_cast<char*>(ws);
    auto fix = [&](std::vector<CBlk>& v) {
        for (CBlk& b : v) {
            b.W = reinterpret_cast<d2*>(base + reinterpret_cast<size_t>(b.W));
            b.Qc = reinterpret_cast<d2*>(base + reinterpret_cast<size_t>(b.Qc));
            b.tau = reinterpret_cast<d2*>(base + reinterpret_cast<size_t>(b.tau));
            b.scale = reinterpret_cast<d2*>(base + reinterpret_cast<size_t>(b.scale));
            b.beta = reinterpret_cast<double*>(base + reinterpret_cast<size_t>(b.beta));
        }
    };
    fix(small);
    fix(large);
    if (!small.empty()) {
        void* dsm = nullptr;
        CYB_TRY(ctx->upload(small.data(), sizeof(CBlk) * small.size(), &dsm));
        hipLaunchKernelGGL(cqh_single_kernel, dim3((unsigned)small.size()), dim3(NT), 0, ctx->stream, static_cast<const CBlk*>(dsm));
        CYB_HIP(hipGetLastError());
    }
    if (!large.empty()) {
        void* dlg = nullptr;
        CYB_TRY(ctx->upload(large.data(), sizeof(CBlk) * large.size(), &dlg));
        const CBlk* dl = static_cast<const CBlk*>(dlg);
        const unsigned nb = (unsigned)large.size();
        int kmax = 0;
        int64_t max_elems = 0;
        for (const CBlk& b : large) {
            kmax = std::max(kmax, b.k);
            max_elems = std::max<int64_t>(max_elems, (int64_t)b.m * std::max(b.n, b.kq));
        }
        const unsigned chunks = (unsigned)std::min<int64_t>(2048, (max_elems + NT - 1) / NT);
        hipLaunchKernelGGL(cqh_load_kernel, dim3(chunks, nb), dim3(NT), 0, ctx->stream, dl);
        for (int j = 0; j < kmax; ++j) {
            int trailing = 0;
            for (const CBlk& b : large)
                if (j < b.k) trailing = std::max(trailing, b.n - j - 1);
            const unsigned gx = (unsigned)std::max(1, std::min(1024, (trailing + NW - 1) / NW));
            hipLaunchKernelGGL(cqh_step_kernel, dim3(gx, nb), dim3(NT), 0, ctx->stream, dl, j);
        }
        for (int j = kmax - 1; j >= 0; --j) {
            int cols = 0;
            for (const CBlk& b : large)
                if (j < b.k) cols = std::max(cols, b.kq - j);
            const unsigned gx = (unsigned)std::max(1, std::min(1024, (cols + NW - 1) / NW));
            hipLaunchKernelGGL(cqh_qstep_kernel, dim3(gx, nb), dim3(NT), 0, ctx->stream, dl, j);
        }
        hipLaunchKernelGGL(cqh_write_kernel, dim3(chunks, nb), dim3(NT), 0, ctx->stream, dl);
        CYB_HIP(hipGetLastError());
    }
    return CYB_OK;
}
