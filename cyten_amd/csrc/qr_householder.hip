// Batched Householder QR (gfx950).
//
// Replaces NumpyBlockBackend::matrix_qr = scipy.linalg.qr(a, mode='economic'|'full')
// (reference src/block_backend/numpy.cpp:1236-1245; LAPACK dgeqrf + dorgqr), called once per
// sector block from AbelianBackend::qr (src/backends/abelian.cpp:3123) and
// FusionTreeBackend::qr (fusion_tree_backend.cpp:2125).  matrix_lq (block_backend.cpp:1033-1040)
// is transpose + this on the host side.
//
// One workgroup per block.  The working copy is kept COLUMN-major so that every Householder
// vector and every column it is applied to is a contiguous run: each wave owns whole columns,
// computes v^T a with a wave reduction (shuffles, no LDS) and updates the column in place.
// Same sign convention as LAPACK dlarfg:  beta = -sign(alpha) * norm,  H = I - tau v v^T, v[0]=1.
#include "blocked_qr.h"

#include <algorithm>
#include <cstdlib>

namespace {

#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS double* gp;
typedef const GLOBAL_AS double* gcp;

constexpr int QNT = 512;
constexpr int QNW = QNT / 64;

struct QrWork {
    const double* A;
    int64_t lda;
    int32_t m, n, k, kq; // k = min(m,n); kq = columns of Q (k or m)
    double* Ac;          // n x m  (column-major working copy of A: Ac[c*m + i] = A[i][c])
    double* Qc;          // kq x m (column-major Q)
    double* tau;         // k
    double* Q;
    int64_t ldq;
    double* R;
    int64_t ldr;
    int32_t r_rows; // rows of R written (k economic, m full)
    int32_t pad;
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(QNT) qr_householder_kernel(const QrWork* __restrict__ works)
{
    __shared__ double red[QNW];
    __shared__ double s_tau, s_scale;
    const QrWork w = works[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = w.m, n = w.n, k = w.k;
    gcp A = (gcp)w.A;
    gp Ac = (gp)w.Ac;
    gp Qc = (gp)w.Qc;
    gp tau = (gp)w.tau;

    // ---- column-major working copy
    for (int64_t e = tid; e < (int64_t)m * n; e += QNT) {
        const int c = (int)(e / m), i = (int)(e % m);
        Ac[e] = A[(int64_t)i * w.lda + c];
    }
    __syncthreads();

    // ---- factorisation
    for (int j = 0; j < k; ++j) {
        gp x = Ac + (int64_t)j * m + j; // x[0..m-j)
        const int L = m - j;
        // norm of x[1:], scaled against overflow by max |x|
        double mx = 0.0;
        for (int i = 1 + tid; i < L; i += QNT) mx = fmax(mx, fabs(x[i]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = 0.0;
#pragma unroll
        for (int q = 0; q < QNW; ++q) mx = fmax(mx, red[q]);
        __syncthreads();
        double ss = 0.0;
        if (mx > 1e-290) { // (a tail this small -- denormal -- is zero: 1/mx would overflow)
            const double inv = 1.0 / mx;
            for (int i = 1 + tid; i < L; i += QNT) {
                const double t = x[i] * inv;
                ss += t * t;
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int q = 0; q < QNW; ++q) tot += red[q];
            const double xnorm = mx * sqrt(tot);
            const double alpha = x[0];
            double t = 0.0, scale = 0.0, beta = alpha;
            if (xnorm != 0.0) {
                beta = -copysign(hypot(alpha, xnorm), alpha);
                t = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            s_tau = t;
            s_scale = scale;
            tau[j] = t;
            x[0] = beta; // R[j][j]
        }
        __syncthreads();
        const double tj = s_tau, scale = s_scale;
        if (tj != 0.0) {
            for (int i = 1 + tid; i < L; i += QNT) x[i] *= scale; // v[1:], v[0] = 1 implicit
            __syncthreads();
            // apply H_j to the trailing columns, one wave per column
            for (int c = j + 1 + wave; c < n; c += QNW) {
                gp a = Ac + (int64_t)c * m + j;
                double dot = (lane == 0) ? a[0] : 0.0;
                for (int i = 1 + lane; i < L; i += 64) dot += x[i] * a[i];
                dot = wave_sum(dot) * tj;
                if (lane == 0) a[0] -= dot;
                for (int i = 1 + lane; i < L; i += 64) a[i] -= dot * x[i];
            }
        }
        __syncthreads();
    }

    // ---- R (upper triangle / trapezoid), row-major output
    for (int64_t e = tid; e < (int64_t)w.r_rows * n; e += QNT) {
        const int i = (int)(e / n), c = (int)(e % n);
        ((gp)w.R)[(int64_t)i * w.ldr + c] = (i <= c && i < k) ? Ac[(int64_t)c * m + i] : 0.0;
    }

    // ---- Q = H_0 H_1 ... H_{k-1} applied to the first kq columns of the identity
    const int kq = w.kq;
    for (int64_t e = tid; e < (int64_t)kq * m; e += QNT) Qc[e] = ((int)(e / m) == (int)(e % m)) ? 1.0 : 0.0;
    __syncthreads();
    for (int j = k - 1; j >= 0; --j) {
        const double tj = tau[j];
        if (tj != 0.0) {
            gcp v = Ac + (int64_t)j * m + j;
            const int L = m - j;
            // columns c < j of the partial product are still unit vectors e_c: untouched by H_j
            for (int c = j + wave; c < kq; c += QNW) {
                gp q = Qc + (int64_t)c * m + j;
                double dot = (lane == 0) ? q[0] : 0.0;
                for (int i = 1 + lane; i < L; i += 64) dot += v[i] * q[i];
                dot = wave_sum(dot) * tj;
                if (lane == 0) q[0] -= dot;
                for (int i = 1 + lane; i < L; i += 64) q[i] -= dot * v[i];
            }
        }
        __syncthreads();
    }
    for (int64_t e = tid; e < (int64_t)m * kq; e += QNT) {
        const int i = (int)(e / kq), c = (int)(e % kq);
        ((gp)w.Q)[(int64_t)i * w.ldq + c] = Qc[(int64_t)c * m + i];
    }
}

} // namespace

// Blocks at least this large in both... dimensions go through the blocked (GEMM-based) path.
static constexpr int64_t kBlockedMin = 96;

static int qr_blocked(cyb_ctx_t ctx, const std::vector<cyb_qr_desc>& ds)
{
    using namespace cyb;
    if (ds.empty()) return CYB_OK;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    std::vector<BqrMat> mats(ds.size());
    std::vector<size_t> oA(ds.size()), oAux(ds.size()), oC(ds.size());
    std::vector<int> kq(ds.size());
    size_t off = 0;
    for (size_t i = 0; i < ds.size(); ++i) {
        const auto& d = ds[i];
        const int64_t k = std::min(d.m, d.n);
        kq[i] = (int)(d.full ? d.m : k);
        oA[i] = off;
        off += al(sizeof(double) * (size_t)d.m * d.n);
        oAux[i] = off;
        off += bqr_aux_bytes(d.m, d.n, d.m, kq[i]);
        oC[i] = off;
        off += al(sizeof(double) * (size_t)d.m * kq[i]);
    }
    void* ws = nullptr;
    CYB_TRY(ctx->workspace(off, &ws, 1));
    char* base = static_cast<char*>(ws);
    std::vector<XposeDesc> x_in, x_r, x_q;
    std::vector<EyeDesc> eyes;
    std::vector<BqrTarget> targets;
    for (size_t i = 0; i < ds.size(); ++i) {
        const auto& d = ds[i];
        BqrMat& q = mats[i];
        q.Ac = reinterpret_cast<double*>(base + oA[i]);
        q.ld = d.m;
        q.m = (int)d.m;
        q.n = (int)d.n;
        q.k = (int)std::min(d.m, d.n);
        bqr_carve(q, base + oAux[i], kq[i]);
        double* Cq = reinterpret_cast<double*>(base + oC[i]);
        // Ac (col-major m x n) seen as a row-major n x m matrix: out(r = column, c = row) = A[c][r]
        x_in.push_back(XposeDesc{d.A, q.Ac, d.lda, d.m, (int)d.n, (int)d.m, 0, 0, 0, 0});
        // R (row-major r_rows x n): out(r, c) = Ac[c*ld + r], zero below the diagonal / beyond row k
        x_r.push_back(XposeDesc{q.Ac, d.R, d.m, d.ldr, d.full ? (int)d.m : q.k, (int)d.n, 1, q.k, 0, 0});
        eyes.push_back(EyeDesc{Cq, d.m, (int)d.m, kq[i], 0, 0});
        targets.push_back(BqrTarget{(int)i, Cq, d.m, kq[i]});
        // Q (row-major m x kq): out(r, c) = Cq[c*m + r]
        x_q.push_back(XposeDesc{Cq, d.Q, d.m, d.ldq, (int)d.m, kq[i], 0, 0, 0, 0});
    }
    CYB_TRY(xpose_batched(ctx, x_in));
    CYB_TRY(bqr_factor(ctx, mats));
    CYB_TRY(xpose_batched(ctx, x_r));
    CYB_TRY(eye_cols_batched(ctx, eyes));
    CYB_TRY(bqr_apply_q(ctx, mats, targets));
    CYB_TRY(xpose_batched(ctx, x_q));
    return CYB_OK;
}

static int qr_batched_impl(cyb_ctx_t ctx, const cyb_qr_desc* descs, int64_t nmat)
{
    CYB_REQUIRE(ctx, "cyb_qr_batched_f64: ctx is NULL");
    CYB_REQUIRE(nmat >= 0 && (nmat == 0 || descs), "cyb_qr_batched_f64: bad descriptor list");
    std::vector<QrWork> works;
    std::vector<size_t> oAc, oQc, oTau;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    std::vector<cyb_qr_desc> big;
    for (int64_t b = 0; b < nmat; ++b) {
        const cyb_qr_desc& d = descs[b];
        CYB_REQUIRE(d.m >= 0 && d.n >= 0 && d.m < (1 << 30) && d.n < (1 << 30), "qr block %lld: bad shape", (long long)b);
        if (d.m == 0 || d.n == 0) continue;
        CYB_REQUIRE(d.A && d.Q && d.R, "qr block %lld: NULL pointer", (long long)b);
        {
            const int64_t kq_ = d.full ? d.m : std::min(d.m, d.n);
            CYB_REQUIRE(d.lda >= d.n && d.ldq >= kq_ && d.ldr >= d.n, "qr block %lld: leading dimension too small", (long long)b);
        }
        if (std::min(d.m, d.n) >= kBlockedMin && getenv("CYB_QR_UNBLOCKED") == nullptr) {
            big.push_back(d);
            continue;
        }
        QrWork w;
        w.A = d.A;
        w.lda = d.lda;
        w.m = (int)d.m;
        w.n = (int)d.n;
        w.k = (int)std::min(d.m, d.n);
        w.kq = d.full ? (int)d.m : w.k;
        w.r_rows = d.full ? (int)d.m : w.k;
        CYB_REQUIRE(d.lda >= d.n && d.ldq >= w.kq && d.ldr >= d.n, "qr block %lld: leading dimension too small", (long long)b);
        w.Q = d.Q;
        w.ldq = d.ldq;
        w.R = d.R;
        w.ldr = d.ldr;
        w.pad = 0;
        w.Ac = w.Qc = w.tau = nullptr;
        oAc.push_back(take(sizeof(double) * (size_t)d.m * d.n));
        oQc.push_back(take(sizeof(double) * (size_t)d.m * w.kq));
        oTau.push_back(take(sizeof(double) * (size_t)w.k));
        works.push_back(w);
    }
    CYB_TRY(qr_blocked(ctx, big));
    if (works.empty()) return CYB_OK;
    void* ws = nullptr;
    CYB_TRY(ctx->workspace(off, &ws));
    char* base = static_cast<char*>(ws);
    for (size_t i = 0; i < works.size(); ++i) {
        works[i].Ac = reinterpret_cast<double*>(base + oAc[i]);
        works[i].Qc = reinterpret_cast<double*>(base + oQc[i]);
        works[i].tau = reinterpret_cast<double*>(base + oTau[i]);
    }
    void* d_w = nullptr;
    CYB_TRY(ctx->upload(works.data(), sizeof(QrWork) * works.size(), &d_w));
    hipLaunchKernelGGL(qr_householder_kernel, dim3((unsigned)works.size()), dim3(QNT), 0, ctx->stream,
                       static_cast<const QrWork*>(d_w));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

// range-safe entry point (scaling.hip): QR of s*A = Q (s*R), s a power of two
extern "C" int cyb_qr_batched_f64(cyb_ctx_t ctx, const cyb_qr_desc* descs, int64_t nmat)
{
    CYB_REQUIRE(ctx, "cyb_qr_batched_f64: ctx is NULL");
    CYB_REQUIRE(nmat >= 0 && (nmat == 0 || descs), "cyb_qr_batched_f64: bad descriptor list");
    std::vector<cyb::MatRef> refs;
    std::vector<int64_t> which;
    for (int64_t b = 0; b < nmat; ++b)
        if (descs[b].m > 0 && descs[b].n > 0 && descs[b].A) {
            refs.push_back(cyb::MatRef{descs[b].A, descs[b].lda, descs[b].m, descs[b].n});
            which.push_back(b);
        }
    std::vector<double> amax;
    CYB_TRY(cyb::matrix_amax(ctx, refs, amax));
    std::vector<cyb_qr_desc> mod;
    std::vector<void*> temps;
    std::vector<cyb::ScaleJob> pre, post;
    for (size_t k = 0; k < refs.size(); ++k) {
        const double sc = cyb::range_scale(amax[k]);
        if (sc == 1.0) continue;
        if (mod.empty()) mod.assign(descs, descs + nmat);
        cyb_qr_desc& d = mod[(size_t)which[k]];
        void* t = nullptr;
        CYB_HIP(hipMalloc(&t, sizeof(double) * (size_t)d.m * (size_t)d.n));
        temps.push_back(t);
        pre.push_back(cyb::ScaleJob{d.A, d.lda, static_cast<double*>(t), d.n, d.m, d.n, sc});
        d.A = static_cast<const double*>(t);
        d.lda = d.n;
        const int64_t r_rows = d.full ? d.m : std::min(d.m, d.n);
        post.push_back(cyb::ScaleJob{d.R, d.ldr, d.R, d.ldr, r_rows, d.n, 1.0 / sc});
    }
    if (mod.empty()) return qr_batched_impl(ctx, descs, nmat);
    int st = cyb::scale_copy_batched(ctx, pre);
    if (st == CYB_OK) st = qr_batched_impl(ctx, mod.data(), nmat);
    if (st == CYB_OK) st = cyb::scale_copy_batched(ctx, post);
    (void)hipStreamSynchronize(ctx->stream);
    for (void* t : temps) (void)hipFree(t);
    return st;
}
