// Batched blocked Householder QR (see blocked_qr.h).  Replaces LAPACK dgeqrf/dorgqr behind
// scipy.linalg.qr (reference src/block_backend/numpy.cpp:1236-1245) for large blocks and is the
// preconditioner / null-space provider of the Jacobi SVD (svd_jacobi.hip).
#include "blocked_qr.h"

#include <algorithm>

namespace cyb {
namespace {

#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS double* gp;
typedef const GLOBAL_AS double* gcp;

constexpr int PNT = 512;
constexpr int PNW = PNT / 64;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct PanelDesc {
    double* Ac;
    double* V;
    double* T;   // this panel's NBK x NBK block
    double* tau;
    int64_t ld;
    int32_t m, j0, pw, pad;
};

// Factor columns [j0, j0+pw) over rows [j0, m): reflectors into V (explicit), R entries stay in Ac,
// tau, and the panel's triangular T factor (dlarft, forward columnwise).
__global__ void __launch_bounds__(PNT) qr_panel_kernel(const PanelDesc* __restrict__ descs)
{
    __shared__ double red[PNW];
    __shared__ double s_tau, s_scale;
    __shared__ double Ts[NBK][NBK + 1];
    __shared__ double z[NBK];
    __shared__ double taus[NBK];
    const PanelDesc d = descs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    gp Ac = (gp)d.Ac;
    gp V = (gp)d.V;
    const int64_t ld = d.ld;
    const int m = d.m, j0 = d.j0, pw = d.pw;
    for (int e = tid; e < NBK * (NBK + 1); e += PNT) (&Ts[0][0])[e] = 0.0;

    for (int jj = 0; jj < pw; ++jj) {
        const int col = j0 + jj;
        gp x = Ac + (int64_t)col * ld + col; // x[0..L)
        const int L = m - col;
        double mx = 0.0;
        for (int i = 1 + tid; i < L; i += PNT) mx = fmax(mx, fabs(x[i]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = 0.0;
#pragma unroll
        for (int q = 0; q < PNW; ++q) mx = fmax(mx, red[q]);
        __syncthreads();
        double ss = 0.0;
        if (mx > 0.0) {
            const double inv = 1.0 / mx;
            for (int i = 1 + tid; i < L; i += PNT) {
                const double t = x[i] * inv;
                ss += t * t;
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int q = 0; q < PNW; ++q) tot += red[q];
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
            taus[jj] = t;
            ((gp)d.tau)[col] = t;
            x[0] = beta; // R[col][col]
        }
        __syncthreads();
        const double tj = s_tau, scale = s_scale;
        // explicit reflector column: zeros above, 1 on the diagonal, scaled tail below
        gp v = V + (int64_t)col * ld;
        for (int i = tid; i < m; i += PNT) {
            double val = 0.0;
            if (i == col) val = 1.0;
            else if (i > col) val = (tj != 0.0) ? x[i - col] * scale : 0.0;
            v[i] = val;
        }
        __syncthreads();
        if (tj != 0.0) {
            // apply H to the remaining panel columns, one wave per column
            for (int c = col + 1 + wave; c < j0 + pw; c += PNW) {
                gp a = Ac + (int64_t)c * ld + col;
                gcp vv = v + col;
                double dot = 0.0;
                for (int i = lane; i < L; i += 64) dot += vv[i] * a[i];
                dot = wave_sum(dot) * tj;
                for (int i = lane; i < L; i += 64) a[i] -= dot * vv[i];
            }
        }
        // z[i] = v_i . v_jj for i < jj (for the T factor)
        for (int i = wave; i < jj; i += PNW) {
            gcp vi = V + (int64_t)(j0 + i) * ld;
            double dot = 0.0;
            for (int r = col + lane; r < m; r += 64) dot += vi[r] * v[r];
            dot = wave_sum(dot);
            if (lane == 0) z[i] = dot;
        }
        __syncthreads();
        // T[0:jj, jj] = -tau_jj * T[0:jj, 0:jj] z ;  T[jj][jj] = tau_jj
        if (tid < jj) {
            double acc = 0.0;
            for (int l = tid; l < jj; ++l) acc += Ts[tid][l] * z[l];
            Ts[tid][jj] = -tj * acc;
        }
        if (tid == 0) Ts[jj][jj] = tj;
        __syncthreads();
    }
    for (int e = tid; e < NBK * NBK; e += PNT) ((gp)d.T)[e] = Ts[e / NBK][e % NBK];
}

inline size_t al256(size_t b) { return (b + 255) / 256 * 256; }

__global__ void __launch_bounds__(256) xpose_kernel(const XposeDesc* __restrict__ descs)
{
    __shared__ double tile[32][33];
    const XposeDesc d = descs[blockIdx.y];
    gcp in = (gcp)d.in;
    gp out = (gp)d.out;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (d.plain) {
        const int64_t tot = (int64_t)d.R * d.C;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (int64_t)gridDim.x * 256) {
            const int64_t r = e / d.C, c = e % d.C;
            out[r * d.ldo + c] = in[r * d.ldi + c];
        }
        return;
    }
    const int tr = (d.R + 31) / 32, tc = (d.C + 31) / 32;
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
        const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { // read in[c*ldi + r]: r contiguous
            const int c = c0 + ty + 8 * q, r = r0 + tx;
            tile[ty + 8 * q][tx] = (c < d.C && r < d.R) ? in[(int64_t)c * d.ldi + r] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { // write out[r*ldo + c]: c contiguous
            const int r = r0 + ty + 8 * q, c = c0 + tx;
            if (r < d.R && c < d.C) {
                double v = tile[tx][ty + 8 * q];
                if (d.upper && (r > c || r >= d.rlim)) v = 0.0;
                out[(int64_t)r * d.ldo + c] = v;
            }
        }
    }
}

__global__ void __launch_bounds__(256) eye_cols_kernel(const EyeDesc* __restrict__ descs)
{
    const EyeDesc d = descs[blockIdx.y];
    gp C = (gp)d.C;
    const int64_t tot = (int64_t)d.m * d.kc;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (int64_t)gridDim.x * 256) {
        const int64_t c = e / d.m, i = e % d.m;
        C[c * d.ld + i] = (i == c + d.col0) ? 1.0 : 0.0;
    }
}

} // namespace

int xpose_batched(cyb_ctx_t ctx, const std::vector<XposeDesc>& descs)
{
    if (descs.empty()) return CYB_OK;
    void* d = nullptr;
    CYB_TRY(ctx->upload(descs.data(), sizeof(XposeDesc) * descs.size(), &d));
    hipLaunchKernelGGL(xpose_kernel, dim3(64, (unsigned)descs.size()), dim3(256), 0, ctx->stream, static_cast<const XposeDesc*>(d));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

int eye_cols_batched(cyb_ctx_t ctx, const std::vector<EyeDesc>& descs)
{
    if (descs.empty()) return CYB_OK;
    void* d = nullptr;
    CYB_TRY(ctx->upload(descs.data(), sizeof(EyeDesc) * descs.size(), &d));
    hipLaunchKernelGGL(eye_cols_kernel, dim3(64, (unsigned)descs.size()), dim3(256), 0, ctx->stream, static_cast<const EyeDesc*>(d));
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}

size_t bqr_aux_bytes(int64_t m, int64_t n, int64_t ld, int64_t kc)
{
    const int64_t k = std::min(m, n);
    const int64_t npan = (k + NBK - 1) / NBK;
    return al256(sizeof(double) * (size_t)ld * (size_t)std::max<int64_t>(k, 1)) + al256(sizeof(double) * (size_t)npan * NBK * NBK) +
           al256(sizeof(double) * (size_t)std::max<int64_t>(k, 1)) + al256(sizeof(double) * 2 * NBK * (size_t)std::max<int64_t>(std::max(n, kc), 1));
}

size_t bqr_carve(BqrMat& q, char* base, int64_t kc)
{
    const int64_t k = q.k;
    const int64_t npan = (k + NBK - 1) / NBK;
    size_t off = 0;
    q.V = reinterpret_cast<double*>(base + off);
    off += al256(sizeof(double) * (size_t)q.ld * (size_t)std::max<int64_t>(k, 1));
    q.T = reinterpret_cast<double*>(base + off);
    off += al256(sizeof(double) * (size_t)npan * NBK * NBK);
    q.tau = reinterpret_cast<double*>(base + off);
    off += al256(sizeof(double) * (size_t)std::max<int64_t>(k, 1));
    q.scratch = reinterpret_cast<double*>(base + off);
    q.scr_half = (int64_t)NBK * std::max<int64_t>(std::max<int64_t>(q.n, kc), 1);
    off += al256(sizeof(double) * 2 * NBK * (size_t)std::max<int64_t>(std::max<int64_t>(q.n, kc), 1));
    return off;
}

int bqr_factor(cyb_ctx_t ctx, const std::vector<BqrMat>& mats)
{
    int max_pan = 0;
    for (const auto& q : mats) max_pan = std::max(max_pan, (q.k + NBK - 1) / NBK);
    for (int p = 0; p < max_pan; ++p) {
        std::vector<PanelDesc> pd;
        GemmBatch g1, g2, g3;
        for (const auto& q : mats) {
            const int j0 = p * NBK;
            if (j0 >= q.k) continue;
            const int pw = std::min(NBK, q.k - j0);
            pd.push_back(PanelDesc{q.Ac, q.V, q.T + (size_t)p * NBK * NBK, q.tau, q.ld, q.m, j0, pw, 0});
            const int j1 = j0 + pw;
            const int64_t nt = q.n - j1, mr = q.m - j0;
            if (nt <= 0) continue;
            double* W1 = q.scratch;
            double* W2 = q.scratch + q.scr_half;
            const double* Vp = q.V + (size_t)j0 * q.ld + j0;        // (i,a) at a*ld + i
            double* At = q.Ac + (size_t)j1 * q.ld + j0;             // (i,c) at c*ld + i
            const double* Tp = q.T + (size_t)p * NBK * NBK;
            // W1 (pw x nt) = Vp^T At
            g1.add(W1, pw, nt, nt, Vp, q.ld, 1, At, 1, q.ld, mr, 1.0, 0.0);
            // W2 (pw x nt) = T^T W1
            g2.add(W2, pw, nt, nt, Tp, 1, NBK, W1, nt, 1, pw, 1.0, 0.0);
            // At^T (nt x mr, ld) -= W2^T Vp^T
            g3.add(At, nt, mr, q.ld, W2, 1, nt, Vp, q.ld, 1, pw, -1.0, 1.0);
        }
        if (pd.empty()) break;
        void* d_pd = nullptr;
        CYB_TRY(ctx->upload(pd.data(), sizeof(PanelDesc) * pd.size(), &d_pd));
        hipLaunchKernelGGL(qr_panel_kernel, dim3((unsigned)pd.size()), dim3(PNT), 0, ctx->stream,
                           static_cast<const PanelDesc*>(d_pd));
        CYB_HIP(hipGetLastError());
        CYB_TRY(g1.launch(ctx));
        CYB_TRY(g2.launch(ctx));
        CYB_TRY(g3.launch(ctx));
    }
    return CYB_OK;
}

int bqr_apply_q(cyb_ctx_t ctx, const std::vector<BqrMat>& mats, const std::vector<BqrTarget>& targets)
{
    int max_pan = 0;
    for (const auto& t : targets) max_pan = std::max(max_pan, (mats[(size_t)t.mat].k + NBK - 1) / NBK);
    for (int p = max_pan - 1; p >= 0; --p) {
        GemmBatch g1, g2, g3;
        for (const auto& t : targets) {
            const BqrMat& q = mats[(size_t)t.mat];
            const int j0 = p * NBK;
            if (j0 >= q.k || t.kc <= 0) continue;
            const int pw = std::min(NBK, q.k - j0);
            const int64_t mr = q.m - j0;
            double* W1 = q.scratch;
            double* W2 = q.scratch + q.scr_half;
            CYB_REQUIRE((int64_t)NBK * t.kc <= q.scr_half, "bqr_apply_q: target wider than the carved scratch");
            const double* Vp = q.V + (size_t)j0 * q.ld + j0;
            double* Ct = t.C + j0; // rows j0.. of every column
            const double* Tp = q.T + (size_t)p * NBK * NBK;
            g1.add(W1, pw, t.kc, t.kc, Vp, q.ld, 1, Ct, 1, t.ldc, mr, 1.0, 0.0);   // W1 = Vp^T C
            g2.add(W2, pw, t.kc, t.kc, Tp, NBK, 1, W1, t.kc, 1, pw, 1.0, 0.0);     // W2 = T W1
            g3.add(Ct, t.kc, mr, t.ldc, W2, 1, t.kc, Vp, q.ld, 1, pw, -1.0, 1.0);  // C^T -= W2^T Vp^T
        }
        if (g1.empty()) continue;
        CYB_TRY(g1.launch(ctx));
        CYB_TRY(g2.launch(ctx));
        CYB_TRY(g3.launch(ctx));
    }
    return CYB_OK;
}

} // namespace cyb
