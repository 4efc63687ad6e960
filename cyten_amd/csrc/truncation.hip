// Device-side selection of the singular values to keep.
//
// Reference: TensorBackend::_truncate_singular_values_selection (src/backends/tensor_backend.cpp:139-242), reached from
// AbelianBackend::truncate_singular_values (src/backends/abelian.cpp:3623-3638), which first pulls ALL singular values
// to the host (:3631) and then uploads per-sector masks again.  Here the list stays on the device: one workgroup sorts
// the values by marginal error (up to 8192 entirely in LDS; up to 65536 in 8192-value chunks through LDS with the
// few long-distance steps of the bitonic network in device memory), applies the same constraints in the same order with the same
// "ignore a constraint that would leave no admissible cut" rule, and writes per sector the ascending positions of the
// kept values -- exactly the index tables cyb_mask_gather_batched_f64 consumes -- plus [err, new_norm] and the kept
// counts (the only bytes the host reads: 16 + 8 * n_sectors).
//
// Order of equal marginal errors: the sort key is (value, original position), i.e. the stable order of the
// restatement in oracle/abelian_ref.py, so masks are bit-identical to the host selection.
#include "common.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int NT = 1024;
constexpr int NMAX = 8192; // values one workgroup sorts in LDS
constexpr int NBIG = 65536; // values the chunked variant handles

struct SDesc {
    const double* S;
    int64_t n, offset; // offset of this sector in the concatenated list
    double w;          // weight of the sector's marginal errors: its quantum dimension (tensor_backend.cpp:158-164), 1 for abelian sectors
};

struct Opts {
    int64_t chi_max, chi_min; // chi_max < 0: no limit
    double degeneracy_tol, trunc_cut, svd_min;
    int32_t has_svd_min, minimize_error;
};

__device__ __forceinline__ bool key_less(double ka, int ia, double kb, int ib) { return ka < kb || (ka == kb && ia < ib); }

// block-wide OR / min / max helpers over NT threads (LDS scratch of NT/64 words)
__device__ int block_any(bool v, int* scratch)
{
    const unsigned long long b = __ballot(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = b != 0;
    __syncthreads();
    int r = 0;
    for (int q = 0; q < NT / 64; ++q) r |= scratch[q];
    return r;
}

// exclusive prefix sums over the NT threads of the workgroup (wave scan by shuffles, then the 16 wave totals)
template <typename T>
__device__ T block_excl_scan(T v, T* wave_tot, T* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    T base = 0, all = 0;
    for (int w = 0; w < NT / 64; ++w) {
        if (w < wave) base += wave_tot[w];
        all += wave_tot[w];
    }
    if (total) *total = all;
    return base + inc - v;
}

// BIG: key / idx / good live in device memory (gkey / gidx / ggood, n2 entries each), LDS stages 8192-value chunks of the sort
template <bool BIG>
__global__ void __launch_bounds__(NT) truncate_select_kernel(const SDesc* __restrict__ descs, int n_sectors, int n, int n2, Opts o,
                                                             double* __restrict__ s_all, // workspace: concatenated S
                                                             int64_t* __restrict__ keep_idx, uint8_t* __restrict__ mask_out,
                                                             double* __restrict__ result, // [err, new_norm, counts...]
                                                             double* __restrict__ gkey, int* __restrict__ gidx,
                                                             unsigned char* __restrict__ ggood)
{
    __shared__ double lkey[NMAX];   // marginal errors, then their running sums
    __shared__ int lidx[NMAX];
    __shared__ unsigned char lgood[BIG ? 1 : NMAX];
    double* key = BIG ? gkey : lkey;
    int* idx = BIG ? gidx : lidx;
    unsigned char* good = BIG ? ggood : lgood;
    __shared__ int scratch[NT / 64];
    __shared__ double dtot[NT / 64];
    __shared__ int s_cut;
    const int tid = threadIdx.x;
    // 1. concatenate (global copy for the S lookups after the sort) and form the keys; pads sort last
    for (int s = 0; s < n_sectors; ++s) {
        const SDesc d = descs[s];
        for (int64_t e = tid; e < d.n; e += NT) {
            const double v = d.S[e];
            s_all[d.offset + e] = v;
            key[d.offset + e] = d.w * (v * v);
            idx[d.offset + e] = (int)(d.offset + e);
        }
    }
    for (int e = n + tid; e < n2; e += NT) {
        key[e] = __builtin_huge_val();
        idx[e] = e;
    }
    __syncthreads();
    // 2. bitonic sort, ascending by (key, position)
    auto cmpx = [&](double* kk, int* ii, int e, int p, bool up) { // one compare-exchange
        const double ka = kk[e], kb = kk[p];
        const int ia = ii[e], ib = ii[p];
        const bool swap = up ? key_less(kb, ib, ka, ia) : key_less(ka, ia, kb, ib);
        if (swap) {
            kk[e] = kb, kk[p] = ka;
            ii[e] = ib, ii[p] = ia;
        }
    };
    if (!BIG) {
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < n2 / 2; t += NT) { // one compare-exchange per thread and step
                    const int e = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    cmpx(key, idx, e, e + j, (e & k) == 0);
                }
                __syncthreads();
            }
        }
    } else {
        // the steps of stage k with distance j < NMAX stay inside aligned NMAX-chunks: those run in LDS, chunk by chunk;
        // the direction of element e is that of the global network, ((base + e) & k) == 0
        auto chunk_steps = [&](int base, int k, int jmax) {
            for (int e = tid; e < NMAX; e += NT) {
                lkey[e] = gkey[base + e];
                lidx[e] = gidx[base + e];
            }
            __syncthreads();
            for (int kk = (jmax == 0 ? 2 : k); kk <= k; kk <<= 1) {
                for (int j = min(kk >> 1, NMAX >> 1); j > 0; j >>= 1) {
                    for (int t = tid; t < NMAX / 2; t += NT) {
                        const int e = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                        cmpx(lkey, lidx, e, e + j, ((base + e) & kk) == 0);
                    }
                    __syncthreads();
                }
            }
            for (int e = tid; e < NMAX; e += NT) {
                gkey[base + e] = lkey[e];
                gidx[base + e] = lidx[e];
            }
            __syncthreads();
        };
        for (int base = 0; base < n2; base += NMAX) chunk_steps(base, NMAX, 0); // all stages k <= NMAX
        for (int k = 2 * NMAX; k <= n2; k <<= 1) {
            for (int j = k >> 1; j >= NMAX; j >>= 1) { // long-distance steps in device memory
                for (int t = tid; t < n2 / 2; t += NT) {
                    const int e = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    cmpx(gkey, gidx, e, e + j, (e & k) == 0);
                }
                __syncthreads();
            }
            for (int base = 0; base < n2; base += NMAX) chunk_steps(base, k, 1); // the steps j < NMAX of stage k
        }
    }
    // 3. admissible cuts.  Each constraint is combined with `good` unless the combination is empty
    //    (combine_constraints of the reference: warn and keep the previous set).
    for (int e = tid; e < n; e += NT) good[e] = 1;
    __syncthreads();
    auto combine = [&](auto g2) {
        bool any = false;
        for (int e = tid; e < n; e += NT) any |= good[e] && g2(e);
        if (block_any(any, scratch)) {
            for (int e = tid; e < n; e += NT) good[e] = good[e] && g2(e);
        }
        __syncthreads();
    };
    if (o.chi_max >= 0 && o.chi_max < n) combine([&](int e) { return e >= n - (int)o.chi_max; });
    if (o.chi_min > 1) combine([&](int e) { return e < n - (int)o.chi_min + 1; });
    if (o.degeneracy_tol > 0.0)
        combine([&](int e) {
            if (e == 0) return true;
            const double a = s_all[idx[e]], b = s_all[idx[e - 1]];
            return log(a <= 1e-100 ? 1e-100 : a) - log(b <= 1e-100 ? 1e-100 : b) >= o.degeneracy_tol;
        });
    if (o.has_svd_min) combine([&](int e) { return s_all[idx[e]] >= o.svd_min; });
    // running sums of the marginal errors (inclusive scan in place: chunk per thread, then the chunk totals)
    {
        const int chunk = (n + NT - 1) / NT;
        const int b0 = min(tid * chunk, n), b1 = min(b0 + chunk, n);
        double run = 0.0;
        for (int e = b0; e < b1; ++e) {
            run += key[e];
            key[e] = run;
        }
        const double base = block_excl_scan<double>(run, dtot, nullptr);
        for (int e = b0; e < b1; ++e) key[e] += base;
        __syncthreads();
    }
    {
        const double tc2 = o.trunc_cut * o.trunc_cut;
        combine([&](int e) { return key[e] > tc2; });
    }
    // 4. the cut: smallest admissible index (keep as many as allowed) or the largest
    {
        int best = o.minimize_error ? n : -1;
        for (int e = tid; e < n; e += NT)
            if (good[e]) best = o.minimize_error ? min(best, e) : max(best, e);
        for (int off = 32; off > 0; off >>= 1) {
            const int other = __shfl_xor(best, off);
            best = o.minimize_error ? min(best, other) : max(best, other);
        }
        if ((tid & 63) == 0) scratch[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            int b = scratch[0];
            for (int q = 1; q < NT / 64; ++q) b = o.minimize_error ? min(b, scratch[q]) : max(b, scratch[q]);
            s_cut = b;
            const double total = key[n - 1];
            const double err = b > 0 ? key[b - 1] : 0.0;
            result[0] = err;
            result[1] = total - err;
        }
        __syncthreads();
    }
    const int cut = s_cut;
    // 5. mask in the original order, then per sector the ascending positions of the kept values
    for (int e = tid; e < n; e += NT) good[idx[e]] = 0; // (re-used as the mask; every position is written twice at most)
    __syncthreads();
    for (int e = cut + tid; e < n; e += NT) good[idx[e]] = 1;
    __syncthreads();
    for (int e = tid; e < n; e += NT) mask_out[e] = good[e];
    int64_t* counts = reinterpret_cast<int64_t*>(result + 2);
    for (int s = 0; s < n_sectors; ++s) {
        const SDesc d = descs[s];
        const int ns = (int)d.n, o0 = (int)d.offset;
        const int chunk = (ns + NT - 1) / NT;
        const int b0 = min(tid * chunk, ns), b1 = min(b0 + chunk, ns);
        int cnt = 0;
        for (int e = b0; e < b1; ++e) cnt += good[o0 + e];
        int kept = 0;
        int w = block_excl_scan<int>(cnt, scratch, &kept);
        if (tid == 0) counts[s] = kept;
        for (int e = b0; e < b1; ++e)
            if (good[o0 + e]) keep_idx[o0 + w++] = e;
        __syncthreads();
    }
}

} // namespace

extern "C" int cyb_truncate_select_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n_sectors, const cyb_trunc_opts* opts,
                                       int64_t* keep_idx_dev, uint8_t* mask_dev, double* result_dev)
{
    return cyb_truncate_select_weighted_f64(ctx, descs, n_sectors, nullptr, opts, keep_idx_dev, mask_dev, result_dev);
}

extern "C" int cyb_truncate_select_weighted_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n_sectors, const double* sector_weights,
                                                const cyb_trunc_opts* opts, int64_t* keep_idx_dev, uint8_t* mask_dev, double* result_dev)
{
    CYB_REQUIRE(ctx && opts && result_dev, "cyb_truncate_select_f64: NULL argument");
    CYB_REQUIRE(n_sectors >= 0 && (n_sectors == 0 || descs), "cyb_truncate_select_f64: bad sector list");
    std::vector<SDesc> hd((size_t)n_sectors);
    int64_t n = 0;
    for (int64_t s = 0; s < n_sectors; ++s) {
        CYB_REQUIRE(descs[s].n >= 0 && (descs[s].n == 0 || descs[s].x), "cyb_truncate_select_f64: sector %lld is malformed", (long long)s);
        const double w = sector_weights ? sector_weights[s] : 1.0;
        CYB_REQUIRE(w > 0.0 && w < 1e300, "cyb_truncate_select_weighted_f64: weight of sector %lld is not a positive finite number", (long long)s);
        hd[(size_t)s] = SDesc{descs[s].x, descs[s].n, n, w};
        n += descs[s].n;
    }
    CYB_REQUIRE(n >= 1, "cyb_truncate_select_f64: no singular values");
    CYB_REQUIRE(keep_idx_dev && mask_dev, "cyb_truncate_select_f64: NULL output");
    if (n > NBIG) {
        cyb::set_error("cyb_truncate_select_f64: %lld values exceed the %d one workgroup sorts", (long long)n, NBIG);
        return CYB_ERR_UNSUPPORTED;
    }
    CYB_REQUIRE(opts->chi_min >= 1, "cyb_truncate_select_f64: chi_min must be >= 1");
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    void *d_descs = nullptr, *ws = nullptr;
    CYB_TRY(ctx->upload(hd.data(), sizeof(SDesc) * hd.size(), &d_descs));
    const bool big = n > NMAX;
    if (big) n2 = std::max(n2, 2 * NMAX);
    // workspace: s_all (n doubles) [+ key (n2 doubles), idx (n2 ints), good (n2 bytes) of the chunked variant]
    const size_t off_key = (sizeof(double) * (size_t)n + 255) & ~(size_t)255;
    const size_t off_idx = off_key + sizeof(double) * (size_t)n2, off_good = off_idx + sizeof(int) * (size_t)n2;
    CYB_TRY(ctx->workspace(big ? off_good + (size_t)n2 : sizeof(double) * (size_t)n, &ws, 2));
    Opts o{opts->chi_max, opts->chi_min, opts->degeneracy_tol, opts->trunc_cut, opts->svd_min, opts->has_svd_min,
           opts->minimize_error};
    char* w8 = static_cast<char*>(ws);
    if (big)
        hipLaunchKernelGGL(truncate_select_kernel<true>, dim3(1), dim3(NT), 0, ctx->stream, static_cast<const SDesc*>(d_descs),
                           (int)n_sectors, (int)n, n2, o, static_cast<double*>(ws), keep_idx_dev, mask_dev, result_dev,
                           reinterpret_cast<double*>(w8 + off_key), reinterpret_cast<int*>(w8 + off_idx),
                           reinterpret_cast<unsigned char*>(w8 + off_good));
    else
        hipLaunchKernelGGL(truncate_select_kernel<false>, dim3(1), dim3(NT), 0, ctx->stream, static_cast<const SDesc*>(d_descs),
                           (int)n_sectors, (int)n, n2, o, static_cast<double*>(ws), keep_idx_dev, mask_dev, result_dev,
                           (double*)nullptr, (int*)nullptr, (unsigned char*)nullptr);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}
