// Range safety of the decompositions.  The Jacobi engine squares vector norms (Gram matrices) and the
// Householder kernels divide by column norms, so inputs whose entries sit near the ends of the double range
// (|a| > 1e90 or < 1e-90) would overflow / underflow where LAPACK rescales internally (dlascl in dgesdd /
// dsyevd / dgeqrf's dlarfg).  The entry points measure max|a| of every block (one small launch + one read of
// n doubles); blocks outside the safe range -- rare -- are decomposed as s*A with s an exact power of two and
// the scale is taken out of S / R / the eigenvalues afterwards.  Blocks inside the range are untouched.
#include "common.h"

#include <algorithm>
#include <cmath>

namespace cyb {
namespace {

#define GLOBAL_AS __attribute__((address_space(1)))
typedef const GLOBAL_AS double* gcp;
typedef GLOBAL_AS double* gp;

struct AmaxDesc {
    const double* A;
    int64_t lda;
    int32_t m, n;
    double* out;
};

__global__ void __launch_bounds__(256) matrix_amax_kernel(const AmaxDesc* __restrict__ descs)
{
    __shared__ double red[4];
    const AmaxDesc d = descs[blockIdx.y];
    gcp A = (gcp)d.A;
    double mx = 0.0;
    bool bad = false;
    // (row by row: a 64-bit division per element made this pass 0.10 ms on the chi=4096 theta list)
    if (d.lda == d.n) {
        const int64_t tot = (int64_t)d.m * d.n;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (int64_t)gridDim.x * 256) {
            const double v = A[e];
            bad = bad || !(fabs(v) <= 1.7e308); // NaN or Inf
            mx = fmax(mx, fabs(v));
        }
    } else {
        for (int r = blockIdx.x; r < d.m; r += gridDim.x) {
            gcp row = A + (int64_t)r * d.lda;
            for (int c = threadIdx.x; c < d.n; c += 256) {
                const double v = row[c];
                bad = bad || !(fabs(v) <= 1.7e308);
                mx = fmax(mx, fabs(v));
            }
        }
    }
    if (bad) mx = __builtin_nan("");
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(mx, o);
        mx = (mx != mx || other != other) ? __builtin_nan("") : fmax(mx, other);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int w = 1; w < 4; ++w) r = (r != r || red[w] != red[w]) ? __builtin_nan("") : fmax(r, red[w]);
        ((gp)d.out)[blockIdx.x] = r; // one partial per workgroup, the host takes the max
    }
}

struct ScaleDesc {
    const double* src;
    int64_t lds;
    double* dst;
    int64_t ldd;
    int32_t rows, cols;
    double s;
};

__global__ void __launch_bounds__(256) scale_copy_kernel(const ScaleDesc* __restrict__ descs)
{
    const ScaleDesc d = descs[blockIdx.y];
    gcp src = (gcp)d.src;
    gp dst = (gp)d.dst;
    const int64_t tot = (int64_t)d.rows * d.cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / d.cols, c = e % d.cols;
        dst[r * d.ldd + c] = src[r * d.lds + c] * d.s;
    }
}

} // namespace

int matrix_amax(cyb_ctx_t ctx, const std::vector<MatRef>& mats, std::vector<double>& amax)
{
    const size_t n = mats.size();
    amax.assign(n, 0.0);
    if (n == 0) return CYB_OK;
    constexpr int kParts = 32; // workgroups per matrix (one workgroup pulls ~20 GB/s: a 1442^2 block alone took 0.8 ms)
    void* d_out = nullptr;
    CYB_TRY(ctx->workspace(sizeof(double) * n * kParts, &d_out, 2));
    std::vector<AmaxDesc> ds(n);
    for (size_t i = 0; i < n; ++i)
        ds[i] = AmaxDesc{mats[i].A, mats[i].lda, (int32_t)mats[i].m, (int32_t)mats[i].n, static_cast<double*>(d_out) + i * kParts};
    void* d_ds = nullptr;
    CYB_TRY(ctx->upload(ds.data(), sizeof(AmaxDesc) * n, &d_ds));
    hipLaunchKernelGGL(matrix_amax_kernel, dim3(kParts, (unsigned)n), dim3(256), 0, ctx->stream, static_cast<const AmaxDesc*>(d_ds));
    CYB_HIP(hipGetLastError());
    std::vector<double> parts(n * kParts);
    CYB_TRY(ctx->d2h(parts.data(), d_out, sizeof(double) * parts.size()));
    for (size_t i = 0; i < n; ++i) {
        double r = 0.0;
        for (int p = 0; p < kParts; ++p) {
            const double v = parts[i * kParts + p];
            r = (r != r || v != v) ? std::nan("") : std::max(r, v);
        }
        amax[i] = r;
    }
    return CYB_OK;
}

double range_scale(double amax)
{
    if (!(amax > 0.0) || !(amax <= 1.7e308)) return 1.0; // zero, NaN, Inf: nothing sensible to do
    if (amax <= 1e90 && amax >= 1e-90) return 1.0;
    int e = 0;
    (void)std::frexp(amax, &e); // amax = f * 2^e, f in [0.5, 1)
    return std::ldexp(1.0, -e);
}

int scale_copy_batched(cyb_ctx_t ctx, const std::vector<ScaleJob>& jobs)
{
    if (jobs.empty()) return CYB_OK;
    std::vector<ScaleDesc> ds(jobs.size());
    for (size_t i = 0; i < jobs.size(); ++i)
        ds[i] = ScaleDesc{jobs[i].src, jobs[i].lds, jobs[i].dst, jobs[i].ldd, (int32_t)jobs[i].rows, (int32_t)jobs[i].cols, jobs[i].s};
    void* d_ds = nullptr;
    CYB_TRY(ctx->upload(ds.data(), sizeof(ScaleDesc) * ds.size(), &d_ds));
    hipLaunchKernelGGL(scale_copy_kernel, dim3(64, (unsigned)ds.size()), dim3(256), 0, ctx->stream, static_cast<const ScaleDesc*>(d_ds));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

} // namespace cyb
