// Batched blocked Householder QR (see blocked_qr.h).  Replaces LAPACK dgeqrf/dorgqr behind
// scipy.linalg.qr (reference src/block_backend/numpy.cpp:1236-1245) for large blocks and is the
// preconditioner / null-space provider of the Jacobi SVD (svd_jacobi.hip).
#include "blocked_qr.h"

#include <algorithm>
#include <cstdlib>
#include <utility>

namespace cyb {
namespace {

#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS double* gp;
typedef const GLOBAL_AS double* gcp;
typedef double d4 __attribute__((ext_vector_type(4)));

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
    int32_t m, j0, pw, pad; // pad: PANEL_* flags
    // early stop (BqrMat::ctl): ctl[0] = 0 while the factorisation runs, else 1 + the first panel step that was skipped;
    // ctl[1] = largest squared Frobenius norm of a trailing block seen so far.  parts[0 .. n_parts): the squared norms of
    // the strips the PREVIOUS step's update wrote, i.e. of this step's trailing block A[j0:, j0:]
    double* ctl = nullptr;
    const double* parts = nullptr;
    double rel2 = 0.0;
    int32_t n_parts = 0, step = 0;
};
constexpr int PANEL_V_ZEROED = 1;       // V was zero-filled by the caller: skip the rows above the panel
constexpr int PANEL_REFLECT_ALWAYS = 2; // BqrMat::reflect_always

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
        if (mx > 1e-290) { // (a tail this small -- denormal -- is zero: 1/mx would overflow)
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
            } else if ((d.pad & PANEL_REFLECT_ALWAYS) && alpha != 0.0) {
                beta = -alpha; // H = I - 2 e e^T (see BqrMat::reflect_always)
                t = 2.0;
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

// ---------------------------------------------------------------------------------------------
// Register-resident panel factorisation (rows of the panel <= RP_NT * RP_RPT).
// Thread t owns rows j0 + t + RP_NT*q (q < RP_RPT) of all NBK panel columns in registers.  Per
// column jj ONE fused pass gives, for every column c, the dot of the current column's tail
// x[1:] with column c's tail: c == jj -> |x[1:]|^2 (the Householder norm), c > jj -> the update
// dots, c < jj -> v_c . v_jj for the T factor.  The 32 dots are reduced with a butterfly
// reduce-scatter (17 shuffles per 16 values instead of 96) and summed over the waves in LDS:
// two barriers per column instead of ~six dependent passes over global memory.
#ifndef CYB_QR_READLANE
#define CYB_QR_READLANE 1
#endif
constexpr int RP_NT = 512;
constexpr int RP_RPT = 3;
constexpr int RP_RPT_FUSED = 2;
constexpr int RP_NW = RP_NT / 64;

__device__ __forceinline__ double dshfl_xor(double v, int m) { return __shfl_xor(v, m); }

#ifndef CYB_QR_REDUCE_DPP
#define CYB_QR_REDUCE_DPP 1
#endif
#if CYB_QR_REDUCE_DPP
// The same reduce-scatter without the LDS crossbar and without selects (the panel kernels are bound by their instruction
// count: ~150 instructions per call in the ds_bpermute form below, ~70 here):
//   * 64 -> 32 and 32 -> 16 lanes: v_permlane32_swap / v_permlane16_swap (gfx950) exchange the upper half (the odd rows) of one
//     register with the lower half (the even rows) of another -- after the swap a plain add IS keep + partner's send;
//   * 16 -> 8 and 8 -> 4 lanes: two DPP moves with complementary bank masks (row_mirror / row_half_mirror: involutions that pair
//     the lower with the upper half) build keep-or-partner in one register and partner-or-keep in the other; their sum is the result;
//   * inside a quad: quad_perm all-reduce.
// Same result lanes and value indices as the form below; the order of the additions differs.
__device__ __forceinline__ void dswap32(double& a, double& b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void dswap16(double& a, double& b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int CTRL, int BANKS>
__device__ __forceinline__ double ddpp(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANKS, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double reduce_scatter16(double (&v)[16], int /*lane*/)
{
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double x = v[i], y = v[i + 8];
        dswap32(x, y); // lanes < 32: (own v[i], partner's v[i]);  lanes >= 32: (partner's v[i+8], own v[i+8])
        a[i] = x + y;
    }
    double b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double x = a[i], y = a[i + 4];
        dswap16(x, y); // even rows: value i of both rows of the pair;  odd rows: value i + 4
        b[i] = x + y;
    }
    constexpr int ROW_MIRROR = 0x140, ROW_HALF_MIRROR = 0x141;
    double c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // lanes 0-7 of a row (banks 0, 1) collect b[i], lanes 8-15 (banks 2, 3) collect b[i + 2]
        const double p = ddpp<ROW_MIRROR, 0xc>(b[i], b[i + 2]);     // 0-7: own b[i]          8-15: partner's b[i + 2]
        const double q = ddpp<ROW_MIRROR, 0x3>(b[i + 2], b[i]);     // 0-7: partner's b[i]    8-15: own b[i + 2]
        c[i] = p + q;
    }
    // lanes with (lane & 4) == 0 (banks 0, 2) collect c[0], the others (banks 1, 3) c[1]
    const double p = ddpp<ROW_HALF_MIRROR, 0xa>(c[0], c[1]);
    const double q = ddpp<ROW_HALF_MIRROR, 0x5>(c[1], c[0]);
    double d = p + q;
    d += ddpp<0x4e, 0xf>(d, d); // quad_perm [2, 3, 0, 1]
    d += ddpp<0xb1, 0xf>(d, d); // quad_perm [1, 0, 3, 2]
    return d;
}
#else
// reduce 16 per-lane values over the 64 lanes; on return lanes with (lane & 3) == 0 hold the total of
// value index ((lane>>5)&1)*8 + ((lane>>4)&1)*4 + ((lane>>3)&1)*2 + ((lane>>2)&1)
__device__ __forceinline__ double reduce_scatter16(double (&v)[16], int lane)
{
    double a[8];
    {
        const bool hi = (lane & 32) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double send = hi ? v[i] : v[i + 8];
            const double keep = hi ? v[i + 8] : v[i];
            a[i] = keep + dshfl_xor(send, 32);
        }
    }
    double b[4];
    {
        const bool hi = (lane & 16) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double send = hi ? a[i] : a[i + 4];
            const double keep = hi ? a[i + 4] : a[i];
            b[i] = keep + dshfl_xor(send, 16);
        }
    }
    double c[2];
    {
        const bool hi = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double send = hi ? b[i] : b[i + 2];
            const double keep = hi ? b[i + 2] : b[i];
            c[i] = keep + dshfl_xor(send, 8);
        }
    }
    double d;
    {
        const bool hi = (lane & 4) != 0;
        const double send = hi ? c[0] : c[1];
        const double keep = hi ? c[1] : c[0];
        d = keep + dshfl_xor(send, 4);
    }
    d += dshfl_xor(d, 2);
    d += dshfl_xor(d, 1);
    return d;
}
#endif

struct PanelShared {
    double wred[2][RP_NW][NBK]; // per-wave partial dots, double buffered by the parity of the step
    // broadcast rows of a column step, double buffered by the parity of the step: a thread that is
    // still in step j reads buffer j & 1 while the fast ones already fill (j + 1) & 1, so a step
    // needs only two barriers
    double wsum[2][NBK];     // dots of column j's tail with every column's tail
    double rowb[2][NBK];     // the pivot row of the panel (all columns)
    double Zs[NBK][NBK + 1]; // Zs[c][j] = v_c . v_j (c < j): input of the T recurrence, built at the end
    double taus[NBK];
    double Ts[NBK][NBK + 1];
};

// One column step with a compile-time column index (every register index is static).
template <int JJ, int RPT, int NT>
__device__ __forceinline__ void panel_step(double (&P)[RPT][NBK], PanelShared& sh, const PanelDesc& d, int tid)
{
    if (JJ >= d.pw) return; // uniform
    const int lane = tid & 63, wave = tid >> 6;
    const int j0 = d.j0;
    const int prow = j0 + JJ; // pivot row
    constexpr int pb = JJ & 1;
    // ---- fused pass: dots of column JJ's tail (rows > prow) with every column's tail
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double part[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) part[c] = 0.0;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            // (row j0 + tid + NT * q lies at or above the pivot row j0 + JJ only for q == 0 and tid <= JJ: said so explicitly,
            //  the kernel is bound by its instruction count -- ~900 per column step and wave -- and the compiler cannot see it)
            const double x = (q == 0 && tid <= JJ) ? 0.0 : P[q][JJ];
#pragma unroll
            for (int c = 0; c < 16; ++c) part[c] += x * P[q][h * 16 + c];
        }
        const double r = reduce_scatter16(part, lane);
        if ((lane & 3) == 0) {
            const int idx = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            sh.wred[pb][wave][h * 16 + idx] = r;
        }
    }
    // the owner of the pivot row publishes it (always thread JJ, its first row)
    if (tid == JJ) {
#pragma unroll
        for (int c = 0; c < NBK; ++c) sh.rowb[pb][c] = P[0][c];
    }
    __syncthreads();
#if CYB_QR_READLANE
    // ONE barrier per column: lane c (mod 32) of EVERY wave sums the per-wave partials of column c itself and reads
    // the pivot row's entry c (nine LDS reads in flight together); the scalars of the reflector then come from lane JJ
    // by v_readlane instead of a second round through LDS
    const int lc = lane & (NBK - 1);
    double ws = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) ws += sh.wred[pb][w][lc];
    const double rb = sh.rowb[pb][lc];
    const double alpha = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rb), JJ), __builtin_amdgcn_readlane(__double2loint(rb), JJ));
    const double xn2_raw = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ws), JJ), __builtin_amdgcn_readlane(__double2loint(ws), JJ));
#else
    if (tid < NBK) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) t += sh.wred[pb][w][tid];
        sh.wsum[pb][tid] = t;
    }
    __syncthreads();
    // ---- reflector (every thread computes the scalars redundantly from LDS)
    const double alpha = sh.rowb[pb][JJ];
    const double xn2_raw = sh.wsum[pb][JJ];
#endif
    // PANEL_REFLECT_ALWAYS: a floor of 1e-40 alpha^2 on the squared tail norm turns the exactly-zero tail into the ordinary
    // case with beta = -alpha, tau = 2, v_tail = 0, and changes nothing else (branch-free: an extra branch here costs the
    // register kernel 28 B of scratch; relative to alpha, so that a zero column keeps tau = 0)
    const double xn2 = fma(alpha * alpha, (d.pad & PANEL_REFLECT_ALWAYS) ? 1e-40 : 0.0, xn2_raw);
    double tau = 0.0, scale = 0.0, beta = alpha;
    // (a squared tail norm in the denormal range is zero for the purpose: 1 / (alpha - beta) would overflow and
    //  the reflector would lose its orthogonality -- an exactly rank-deficient block, e.g. all ones, gets there
    //  because every elimination step leaves a trailing block 1e-16 times smaller than the one before)
    if (xn2 > 1e-290) {
        beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    // ---- update my rows: v = x*scale below the pivot, 1 on it, 0 above; columns c > JJ get H applied.
    //      Column-outer order: the broadcast factor of a column is read from LDS once, not per row.
    double v[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const double x = P[q][JJ];
        if (q == 0) { // (only a thread's first row can be the pivot row or lie above it)
            const bool below = tid > JJ;
            const bool pivot = tid == JJ;
            v[q] = below ? x * scale : (pivot ? 1.0 : 0.0);
            P[q][JJ] = below ? v[q] : (pivot ? beta : x);
        } else {
            v[q] = x * scale;
            P[q][JJ] = v[q];
        }
    }
#if CYB_QR_READLANE
    // every lane c < 32 forms the factor of column c once (ONE round trip to LDS for all columns); the update loop
    // then takes it from lane c with v_readlane -- a scalar operand, no LDS latency per column
    const double tl = tau * (rb + scale * ws);
    const int tlo = __double2loint(tl), thi = __double2hiint(tl);
#pragma unroll
    for (int c = JJ + 1; c < NBK; ++c) {
        const double t = __hiloint2double(__builtin_amdgcn_readlane(thi, c), __builtin_amdgcn_readlane(tlo, c));
#pragma unroll
        for (int q = 0; q < RPT; ++q) P[q][c] -= t * v[q]; // (columns beyond pw hold zeros and zero factors)
    }
#else
#pragma unroll
    for (int c = JJ + 1; c < NBK; ++c) {
        // v . a_c = a_c[pivot] + scale * (x_tail . a_c_tail); columns beyond pw hold zeros
        // (the uniform guard also keeps the scheduler from hoisting every column's broadcast loads at
        //  once: without it the kernel spills -- the panel already fills the register file)
        if (c < d.pw) {
            const double t = tau * (sh.rowb[pb][c] + scale * sh.wsum[pb][c]);
#pragma unroll
            for (int q = 0; q < RPT; ++q) P[q][c] -= t * v[q];
        }
    }
#endif
    // ---- inputs of the T factor: z_c = v_c . v_JJ = v_c[pivot] + scale * (v_c_tail . x_tail), c < JJ
#if CYB_QR_READLANE
    if (tid < JJ) sh.Zs[tid][JJ] = rb + scale * ws; // (tid < 32: lane tid holds column tid)
#else
    if (tid < JJ) sh.Zs[tid][JJ] = sh.rowb[pb][tid] + scale * sh.wsum[pb][tid];
#endif
    if (tid == 0) {
        sh.taus[JJ] = tau;
        ((gp)d.tau)[prow] = tau;
    }
    // no barrier here: the broadcast rows are double buffered
}

template <int RPT, int NT, int... Is>
__device__ __forceinline__ void panel_steps(double (&P)[RPT][NBK], PanelShared& sh, const PanelDesc& d, int tid,
                                            std::integer_sequence<int, Is...>)
{
    (panel_step<Is, RPT, NT>(P, sh, d, tid), ...);
}

// RPT rows per thread: 3 for the stand-alone kernel (panels of up to 1536 rows), 2 inside the strip kernel (up to 1024
// rows: 64 registers less, so that the panel body fits behind the strip phases without spilling)
template <int RPT, int NT = RP_NT>
__device__ __forceinline__ void panel_reg_body(const PanelDesc& d, PanelShared& sh, const int tid)
{
    gp Ac = (gp)d.Ac;
    gp V = (gp)d.V;
    const int64_t ld = d.ld;
    const int m = d.m, j0 = d.j0, pw = d.pw;
    // ---- load my rows of the panel.  Branch-free: rows beyond m / columns beyond pw load a valid element (row j0 / column
    //      pw - 1) and are zeroed by a select -- the guarded form cost 18 instructions per element, and these kernels are bound
    //      by their instruction count
    double P[RPT][NBK];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int row = j0 + tid + NT * q;
        const bool ok = row < m;
        gcp src = (gcp)Ac + (int64_t)j0 * ld + (ok ? row : j0);
#pragma unroll
        for (int c = 0; c < NBK; ++c) P[q][c] = src[(int64_t)(c < pw ? c : pw - 1) * ld];
#pragma unroll
        for (int c = 0; c < NBK; ++c) {
            asm volatile("" : "+v"(P[q][c])); // (keeps the loads unconditional and in flight together: the compiler would sink each under its select again)
            P[q][c] = (ok && c < pw) ? P[q][c] : 0.0;
        }
    }
    for (int e = tid; e < NBK * (NBK + 1); e += NT) (&sh.Zs[0][0])[e] = 0.0;
    __syncthreads();

    panel_steps<RPT, NT>(P, sh, d, tid, std::make_integer_sequence<int, NBK>{});
    // ---- write back: Ac (R above / on the diagonal, v below) and the explicit V (1 on the diagonal, 0 above: only a
    //      thread's first row, j0 + tid with tid < NBK, can be on or above the diagonal of the panel)
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int row = j0 + tid + NT * q;
        if (row < m) {
            gp pa = Ac + (int64_t)j0 * ld + row;
            gp pv = V + (int64_t)j0 * ld + row;
#pragma unroll
            for (int c = 0; c < NBK; ++c)
                if (c < pw) {
                    pa[(int64_t)c * ld] = P[q][c];
                    pv[(int64_t)c * ld] = (q > 0 || tid > c) ? P[q][c] : (tid == c ? 1.0 : 0.0);
                }
        }
    }
    // rows above the panel's first row are zero in V
    for (int64_t e = tid; e < ((d.pad & PANEL_V_ZEROED) ? 0 : (int64_t)j0 * pw); e += NT) { // (d.pad: V was zero-filled by the caller)
        const int c = (int)(e / j0), row = (int)(e % j0);
        V[(int64_t)(j0 + c) * ld + row] = 0.0;
    }
    __syncthreads();
    // ---- T factor (compact WY, T upper triangular): column j = -tau_j T[:, :j] z_j.  Row i of T only
    //      needs row i of the earlier columns, so lane i builds its own row with no exchange.
    if (tid < NBK) {
        double trow[NBK];
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < j; ++l) acc += trow[l] * sh.Zs[l][j]; // (trow[l] is zero for l < tid: T is upper triangular)
            const double tj = j < pw ? sh.taus[j] : 0.0;
            trow[j] = (tid < j) ? -tj * acc : (tid == j ? tj : 0.0);
            sh.Ts[tid][j] = trow[j];
        }
    }
    __syncthreads();

    for (int e = tid; e < NBK * NBK; e += NT) ((gp)d.T)[e] = sh.Ts[e / NBK][e % NBK];
}

// Early stop of a factorisation whose trailing block has fallen to the rounding level of the matrix (a rank-deficient block:
// every block of a two-site theta = A.B): the remaining panels would factor noise.  The panel kernel of step p sums the
// squared norms of the strips step p - 1 wrote (fixed order: deterministic) -- that is ||A[j0:, j0:]||_F^2 -- and, once it
// is below rel2 x the largest such norm seen, marks the matrix stopped: this and all later panel and strip launches of the
// matrix return at once, T stays zero (the caller's zeroed workspace), so those reflectors are the identity.
// Returns true if the workgroup is to leave.  `red`: NT / 64 doubles of LDS.
template <int NT, typename Desc>
__device__ __forceinline__ bool panel_stop_check(const Desc& d, double* red, int tid, bool writer = true)
{
    if (!d.ctl) return false;
    gp ctl = (gp)d.ctl;
    const double stopped = ctl[0];
    if (stopped != 0.0) return (double)d.step >= stopped - 1.0;
    if (d.n_parts <= 0) return false;
    gcp parts = (gcp)d.parts;
    double t = 0.0;
    for (int i = tid; i < d.n_parts; i += NT) t += parts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((tid & 63) == 0) red[tid >> 6] = t;
    __syncthreads();
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) sum += red[w];
    const double ref = fmax(ctl[1], sum);
    const bool stop = sum <= d.rel2 * ref;
    __syncthreads(); // (everybody has read ctl before it changes)
    if (tid == 0 && writer) {
        if (stop) ctl[0] = (double)d.step + 1.0;
        else ctl[1] = ref;
    }
    return stop;
}

__global__ void __launch_bounds__(RP_NT) qr_panel_reg_kernel(const PanelDesc* __restrict__ descs)
{
    __shared__ PanelShared sh;
    const PanelDesc d = descs[blockIdx.x];
    if (panel_stop_check<RP_NT>(d, &sh.wsum[0][0], (int)threadIdx.x)) return;
    panel_reg_body<RP_RPT>(d, sh, (int)threadIdx.x);
}
// ---------------------------------------------------------------------------------------------
// Panels of more than RP_NT * RP_RPT = 1536 rows: the register-resident factorisation spread over several workgroups
// (workgroup w owns rows j0 + 1536 w ...), which exchange their partial column dots once per column step through device
// memory -- write-through stores, one ticket per column, a bounded poll by one lane -- instead of the global-memory
// panel kernel (six dependent passes over the panel per column: 1.7 ms per panel at 2884 rows).  The pivot rows of a panel
// (j0 .. j0 + 31) always belong to workgroup 0, which sends the pivot row along with its dots.
struct PanelDescM {
    double* Ac;
    double* V;
    double* T;
    double* tau;
    double* xchg;         // [NBK column steps][n_wg][64]: 32 partial dots (+ the pivot row from workgroup 0)
    unsigned int* ticket; // [NBK], zeroed before the launch
    unsigned int* err;    // set if a poll ran out of time (partner workgroup not resident)
    int64_t ld;
    int32_t m, j0, pw, vzero;
    int32_t wg, n_wg, pad0, pad1;
    // early stop, as in PanelDesc (every workgroup of the matrix sums the same numbers in the same order and takes the same
    // decision; workgroup 0 records it)
    double* ctl = nullptr;
    const double* parts = nullptr;
    double rel2 = 0.0;
    int32_t n_parts = 0, step = 0;
};
constexpr int RPM_ROWS = RP_NT * RP_RPT;

template <int JJ>
__device__ __forceinline__ void panel_step_multi(double (&P)[RP_RPT][NBK], PanelShared& sh, const PanelDescM& d, int tid)
{
    if (JJ >= d.pw) return; // uniform
    const int lane = tid & 63, wave = tid >> 6;
    const int j0 = d.j0, rbase = d.j0 + d.wg * RPM_ROWS;
    const int prow = j0 + JJ;
    constexpr int pb = JJ & 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double part[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) part[c] = 0.0;
#pragma unroll
        for (int q = 0; q < RP_RPT; ++q) {
            const int row = rbase + tid + RP_NT * q;
            const double x = (row <= prow) ? 0.0 : P[q][JJ];
#pragma unroll
            for (int c = 0; c < 16; ++c) part[c] += x * P[q][h * 16 + c];
        }
        const double r = reduce_scatter16(part, lane);
        if ((lane & 3) == 0) {
            const int idx = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            sh.wred[pb][wave][h * 16 + idx] = r;
        }
    }
#pragma unroll
    for (int q = 0; q < RP_RPT; ++q)
        if (rbase + tid + RP_NT * q == prow) {
#pragma unroll
            for (int c = 0; c < NBK; ++c) sh.rowb[pb][c] = P[q][c];
        }
    __syncthreads();
    if (tid < NBK) { // (one wave: the stores of these lanes, one drain, one ticket, one poll, the loads)
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < RP_NW; ++w) t += sh.wred[pb][w][tid];
        double* slot = d.xchg + ((size_t)JJ * d.n_wg + d.wg) * 64;
        __hip_atomic_store(slot + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d.wg == 0) __hip_atomic_store(slot + 32 + tid, sh.rowb[pb][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            // (relaxed on purpose: the data went out as write-through stores that this wave has drained, and comes in through
            //  agent-scope loads that bypass the L2 -- a release / acquire pair here would write back and invalidate the whole
            //  L2 of the XCD on every column step: 1.7 ms per panel instead of 0.3)
            __hip_atomic_fetch_add(d.ticket + JJ, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(d.ticket + JJ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)d.n_wg) {
                if (wall_clock64() - t0 > 100000000ull || __hip_atomic_load(d.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { // 1 s at 100 MHz
                    __hip_atomic_store(d.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        // compiler-only barrier: the relaxed loads below must not be hoisted above the poll of lane 0 (nothing in the C++
        // memory model orders relaxed accesses to different addresses; costs no instruction and no L2 write-back)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double tot = 0.0;
        for (int w = 0; w < d.n_wg; ++w)
            tot += __hip_atomic_load(d.xchg + ((size_t)JJ * d.n_wg + w) * 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh.wsum[pb][tid] = tot;
        sh.rowb[pb][tid] = __hip_atomic_load(d.xchg + (size_t)JJ * d.n_wg * 64 + 32 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const double alpha = sh.rowb[pb][JJ];
    const double xn2 = fma(alpha * alpha, (d.vzero & PANEL_REFLECT_ALWAYS) ? 1e-40 : 0.0, sh.wsum[pb][JJ]);
    double tau = 0.0, scale = 0.0, beta = alpha;
    if (xn2 > 1e-290) {
        beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    double v[RP_RPT];
#pragma unroll
    for (int q = 0; q < RP_RPT; ++q) {
        const int row = rbase + tid + RP_NT * q;
        const double x = P[q][JJ];
        const bool below = row > prow;
        const bool pivot = row == prow;
        v[q] = below ? x * scale : (pivot ? 1.0 : 0.0);
        P[q][JJ] = below ? v[q] : (pivot ? beta : x);
    }
#pragma unroll
    for (int c = JJ + 1; c < NBK; ++c) {
        if (c < d.pw) {
            const double t = tau * (sh.rowb[pb][c] + scale * sh.wsum[pb][c]);
#pragma unroll
            for (int q = 0; q < RP_RPT; ++q) P[q][c] -= t * v[q];
        }
    }
    if (tid < JJ) sh.Zs[tid][JJ] = sh.rowb[pb][tid] + scale * sh.wsum[pb][tid];
    if (tid == 0) {
        sh.taus[JJ] = tau;
        if (d.wg == 0) ((gp)d.tau)[prow] = tau;
    }
}

template <int... Is>
__device__ __forceinline__ void panel_steps_multi(double (&P)[RP_RPT][NBK], PanelShared& sh, const PanelDescM& d, int tid,
                                                  std::integer_sequence<int, Is...>)
{
    (panel_step_multi<Is>(P, sh, d, tid), ...);
}

__global__ void __launch_bounds__(RP_NT) qr_panel_multi_kernel(const PanelDescM* __restrict__ descs)
{
    __shared__ PanelShared sh;
    const PanelDescM d = descs[blockIdx.x];
    const int tid = threadIdx.x;
    if (panel_stop_check<RP_NT>(d, &sh.wsum[0][0], tid, d.wg == 0)) return;
    gp Ac = (gp)d.Ac;
    gp V = (gp)d.V;
    const int64_t ld = d.ld;
    const int m = d.m, j0 = d.j0, pw = d.pw, rbase = d.j0 + d.wg * RPM_ROWS;
    double P[RP_RPT][NBK];
#pragma unroll
    for (int q = 0; q < RP_RPT; ++q) {
        const int row = rbase + tid + RP_NT * q;
#pragma unroll
        for (int c = 0; c < NBK; ++c) P[q][c] = (row < m && c < pw) ? Ac[(int64_t)(j0 + c) * ld + row] : 0.0;
    }
    for (int e = tid; e < NBK * (NBK + 1); e += RP_NT) (&sh.Zs[0][0])[e] = 0.0;
    __syncthreads();
    panel_steps_multi(P, sh, d, tid, std::make_integer_sequence<int, NBK>{});
#pragma unroll
    for (int q = 0; q < RP_RPT; ++q) {
        const int row = rbase + tid + RP_NT * q;
        if (row < m) {
#pragma unroll
            for (int c = 0; c < NBK; ++c)
                if (c < pw) {
                    const int col = j0 + c;
                    Ac[(int64_t)col * ld + row] = P[q][c];
                    V[(int64_t)col * ld + row] = (row > col) ? P[q][c] : (row == col ? 1.0 : 0.0);
                }
        }
    }
    if (d.wg == 0 && !(d.vzero & PANEL_V_ZEROED))
        for (int64_t e = tid; e < (int64_t)j0 * pw; e += RP_NT) {
            const int c = (int)(e / j0), row = (int)(e % j0);
            V[(int64_t)(j0 + c) * ld + row] = 0.0;
        }
    __syncthreads();
    if (tid < NBK) {
        double trow[NBK];
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < j; ++l) acc += trow[l] * sh.Zs[l][j]; // (trow[l] is zero for l < tid: T is upper triangular)
            const double tj = j < pw ? sh.taus[j] : 0.0;
            trow[j] = (tid < j) ? -tj * acc : (tid == j ? tj : 0.0);
            sh.Ts[tid][j] = trow[j];
        }
    }
    __syncthreads();
    if (d.wg == 0)
        for (int e = tid; e < NBK * NBK; e += RP_NT) ((gp)d.T)[e] = sh.Ts[e / NBK][e % NBK];
}

// Panels of at most RP_WAVE_ROWS rows (every panel of a DMRG-sized block, the last panels of a large one): the same
// body on ONE wave -- the per-column reduction over eight waves through LDS and the workgroup barrier behind it, which
// is most of a column step at these sizes, shrink to a wave-local exchange.
constexpr int RP_WAVE_RPT = 3;
constexpr int RP_WAVE_ROWS = 64 * RP_WAVE_RPT;
__global__ void __launch_bounds__(64) qr_panel_wave_kernel(const PanelDesc* __restrict__ descs)
{
    __shared__ PanelShared sh;
    const PanelDesc d = descs[blockIdx.x];
    if (panel_stop_check<64>(d, &sh.wsum[0][0], (int)threadIdx.x)) return;
    panel_reg_body<RP_WAVE_RPT, 64>(d, sh, (int)threadIdx.x);
}
// ... and on two / four waves for up to 2 / 4 * RP_WAVE_ROWS rows (the sectors of a chi = 512 bond; the later panels of a
// large block): the fewer waves take part in the per-column reduction and barrier, the shorter the column step.
// (Tall panels on four waves with six rows per thread were measured too: the 192 panel doubles per thread spill --
//  228 B of scratch -- and the chi=4096 step goes from 36.7 to 40.3 ms; four rows per thread for panels up to 1024 rows:
//  neutral, 36.5 / 36.6 vs 36.6 / 36.3 ms.  Panels above 768 rows keep eight waves x 3 rows.)
__global__ void __launch_bounds__(128) qr_panel_wave2_kernel(const PanelDesc* __restrict__ descs)
{
    __shared__ PanelShared sh;
    const PanelDesc d = descs[blockIdx.x];
    if (panel_stop_check<128>(d, &sh.wsum[0][0], (int)threadIdx.x)) return;
    panel_reg_body<RP_WAVE_RPT, 128>(d, sh, (int)threadIdx.x);
}
__global__ void __launch_bounds__(256) qr_panel_wave4_kernel(const PanelDesc* __restrict__ descs)
{
    __shared__ PanelShared sh;
    const PanelDesc d = descs[blockIdx.x];
    if (panel_stop_check<256>(d, &sh.wsum[0][0], (int)threadIdx.x)) return;
    panel_reg_body<RP_WAVE_RPT, 256>(d, sh, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Block-reflector application to one 32-column strip, ONE launch per panel step for the whole batch:
//     C_s <- (I - V P V^T) C_s,   P = T (apply Q) or T^T (apply Q^T, the trailing update of the factorisation).
// A strip is private to its workgroup, so the three products of the compact-WY update chain inside the
// kernel instead of across launches (the grouped-GEMM formulation needs two launches per panel step,
// ~34 us each at the chi=4096 sizes, almost all of it launch + tail latency of narrow problems):
//   1. W1 = V^T C_s   (32 x 32, K = rows): f64 MFMA 16x16x4, K split over the eight waves, both operands are
//      read as row PAIRS of a column (the two k-steps of a pair use the same pairing on both sides);
//   2. W2 = P W1      (32 x 32 x 32) from LDS, one 16 x 16 tile per wave;
//   3. C_s^T -= W2^T V^T, 16 rows of C_s per MFMA tile in the TRANSPOSED form, so that the lanes of a
//      load/store run along the rows of the column-major strip (coalesced), W2 stays in registers.
// Algorithmic traffic: C_s read twice (the second time from L2/MALL) and written once, V read twice.
struct StripDesc {
    double* C;       // strip: element (i, c) at C[c*ldc + i], i < mr, c < nc
    const double* V; // reflector panel: (i, a) at V[a*ldv + i], i < mr, a < pw
    const double* T; // NBK x NBK row-major, zero outside the leading pw x pw block
    int64_t ldc, ldv;
    int32_t mr, nc, pw, transT;
    int64_t next_off; // factorisation only: this strip holds the columns of the NEXT panel; byte offset of its PanelDesc from the strip descriptors (0: none)
    // early stop (see panel_stop_check): a strip of panel step `step` returns at once if the matrix was stopped at or before
    // that step; a factorisation strip leaves the squared norm of the rows it wrote BELOW the panel's pw rows in *part_out
    const double* ctl = nullptr;
    double* part_out = nullptr;
    int32_t step = 0, pad2 = 0;
};
constexpr int ST_NT_CHECK = 512;
static_assert(ST_NT_CHECK == RP_NT, "the fused look-ahead runs the panel body with the strip kernel's workgroup");
constexpr int ST_NT = 512; // two waves per SIMD: the loads of one hide behind the MFMAs of the other
constexpr int ST_NW = ST_NT / 64;
constexpr int ST_LS = NBK + 1;
constexpr int ST_UN = 4;

// NORM: the strips also leave the squared norm of what they wrote (early stop; a separate instantiation -- the accumulator
// costs the kernel 20 B of scratch, which the plain strips must not pay)
// LA (role-split look-ahead, factorisation only): the launch carries n_strips strip workgroups and, behind them, one PANEL
// workgroup per entry of pnext.  The strip whose columns are the next panel (next_off = index + 1 of that entry) raises
// la_flags[index] to la_tag once its stores are out (agent-scope release); the panel workgroup waits for it (acquire) and
// factors panel p + 1 while the other strips of step p are still running: a panel step costs max(strips, strip 0 + panel)
// instead of their sum, with no cross-stream dependency.  The panel workgroups come LAST in the grid, so every strip they
// wait for has been dispatched before them; a wait that runs out of time (1 s) sets la_flags[n_flags - 1] (read back by the host).
template <bool FUSE_PANEL, bool NORM = false, bool LA = false>
__global__ void __launch_bounds__(ST_NT) reflector_strip_kernel(const StripDesc* __restrict__ descs, int n_strips, const PanelDesc* __restrict__ pnext,
                                                                unsigned int* la_flags, unsigned int la_tag, int la_err)
{
    if constexpr (LA) {
        if ((int)blockIdx.x >= n_strips) {
            __shared__ PanelShared psh;
            const int idx = (int)blockIdx.x - n_strips;
            const PanelDesc pd = pnext[idx];
            if (threadIdx.x == 0) {
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(la_flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != la_tag) {
                    __builtin_amdgcn_s_sleep(8);
                    if (wall_clock64() - t0 > 100000000ll) { // 1 s at 100 MHz
                        __hip_atomic_store(la_flags + la_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (panel_stop_check<RP_NT>(pd, &psh.wsum[0][0], (int)threadIdx.x)) return;
            panel_reg_body<RP_RPT>(pd, psh, (int)threadIdx.x);
            return;
        }
    }
    __shared__ double part[ST_NW][NBK][ST_LS];
    __shared__ double W1s[NBK][ST_LS], W2s[NBK][ST_LS], Ps[NBK][ST_LS];
    const StripDesc d = descs[blockIdx.x];
    if (d.ctl) { // (workgroup-uniform)
        const double stopped = *(gcp)d.ctl;
        if (stopped != 0.0 && (double)d.step >= stopped - 1.0) {
            if constexpr (LA)
                if (d.next_off && threadIdx.x == 0) // (its panel workgroup leaves on the same test)
                    __hip_atomic_store(la_flags + (d.next_off - 1), la_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 15, kq = lane >> 4;
    gcp V = (gcp)d.V;
    gp C = (gp)d.C;
    gcp T = (gcp)d.T;
    const int mr = d.mr, nc = d.nc, pw = d.pw;
    const int64_t ldv = d.ldv, ldc = d.ldc;
    for (int e = tid; e < NBK * NBK; e += ST_NT) {
        const int a = e / NBK, b = e % NBK;
        Ps[a][b] = T[d.transT ? b * NBK + a : e];
    }
    // ---- phase 1: partial W1 of this wave's row chunks (8 rows per step).  Columns beyond pw / nc are
    //      clamped to column 0: their products meet zero rows/columns of P or are never stored.
    {
        const int64_t vo0 = (int64_t)(x < pw ? x : 0) * ldv, vo1 = (int64_t)(x + 16 < pw ? x + 16 : 0) * ldv;
        const int64_t co0 = (int64_t)(x < nc ? x : 0) * ldc, co1 = (int64_t)(x + 16 < nc ? x + 16 : 0) * ldc;
        d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        const int nstep = (mr + 7) / 8;
        double va[ST_UN][2][2], ca[ST_UN][2][2], vn[ST_UN][2][2], cn[ST_UN][2][2]; // [step][tile][row of the pair]
        auto load = [&](double (&v)[ST_UN][2][2], double (&c)[ST_UN][2][2], int s0) {
#pragma unroll
            for (int u = 0; u < ST_UN; ++u) {
                const int i = (s0 + ST_NW * u) * 8 + 2 * kq;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool ok = i + h < mr;
                    const int ii = ok ? i + h : 0;
                    v[u][0][h] = ok ? V[vo0 + ii] : 0.0;
                    v[u][1][h] = ok ? V[vo1 + ii] : 0.0;
                    c[u][0][h] = ok ? C[co0 + ii] : 0.0;
                    c[u][1][h] = ok ? C[co1 + ii] : 0.0;
                }
            }
        };
        if (wave < nstep) load(va, ca, wave);
        for (int s0 = wave; s0 < nstep; s0 += ST_NW * ST_UN) {
            const bool more = s0 + ST_NW * ST_UN < nstep;
            if (more) load(vn, cn, s0 + ST_NW * ST_UN);
#pragma unroll
            for (int u = 0; u < ST_UN; ++u) {
                if (s0 + ST_NW * u < nstep) { // wave-uniform (rows beyond mr inside a step are loaded as zeros)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u][i][h], ca[u][j][h], acc[i][j], 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < ST_UN; ++u)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            va[u][i][h] = vn[u][i][h];
                            ca[u][i][h] = cn[u][i][h];
                        }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave][i * 16 + kq + 4 * r][j * 16 + x] = acc[i][j][r];
    }
    __syncthreads();
    for (int e = tid; e < NBK * NBK; e += ST_NT) {
        const int a = e / NBK, c = e % NBK;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < ST_NW; ++w) t += part[w][a][c];
        W1s[a][c] = t;
    }
    __syncthreads();
    // ---- phase 2: W2 = P W1, tile (wave >> 1, wave & 1)
    if (wave < 4) {
        const int ta = wave >> 1, tc = wave & 1;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NBK / 4; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ps[ta * 16 + x][4 * kk + kq], W1s[4 * kk + kq][tc * 16 + x], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) W2s[ta * 16 + kq + 4 * r][tc * 16 + x] = acc[r];
    }
    __syncthreads();
    // ---- phase 3: (C_s)^T tile [c][i] -= sum_a W2[a][c] V[i][a], 16 rows i per tile, tiles dealt to the waves
    {
        double w2[2][NBK / 4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kk = 0; kk < NBK / 4; ++kk) w2[ct][kk] = -W2s[4 * kk + kq][ct * 16 + x];
        int64_t vcol[NBK / 4], ccol[2][4];
#pragma unroll
        for (int kk = 0; kk < NBK / 4; ++kk) vcol[kk] = (int64_t)(4 * kk + kq < pw ? 4 * kk + kq : 0) * ldv;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = ct * 16 + kq + 4 * r;
                ccol[ct][r] = (int64_t)(c < nc ? c : 0) * ldc;
            }
        const int ntile = (mr + 15) / 16;
        struct Tile {
            double v[NBK / 4];
            d4 c[2];
        };
        auto load = [&](Tile& t, int rt) {
            const int i = rt * 16 + x;
            const bool ok = i < mr;
            const int ii = ok ? i : 0;
#pragma unroll
            for (int kk = 0; kk < NBK / 4; ++kk) t.v[kk] = ok ? V[vcol[kk] + ii] : 0.0;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) t.c[ct][r] = C[ccol[ct][r] + ii];
        };
        double nrm2 = 0.0; // squared norm of the entries this lane writes below the panel's rows
        auto finish = [&](Tile& t, int rt) {
#pragma unroll
            for (int kk = 0; kk < NBK / 4; ++kk) {
                t.c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[0][kk], t.v[kk], t.c[0], 0, 0, 0);
                t.c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[1][kk], t.v[kk], t.c[1], 0, 0, 0);
            }
            const int i = rt * 16 + x;
            if (i < mr) {
                const double below = i >= pw ? 1.0 : 0.0;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ct * 16 + kq + 4 * r < nc) {
                            C[ccol[ct][r] + i] = t.c[ct][r];
                            if constexpr (NORM) nrm2 = fma(below * t.c[ct][r], t.c[ct][r], nrm2);
                        }
            }
        };
        // three tiles in rotation: two are in flight while one is multiplied (a tile is ~0.4 us of MFMA, a miss ~1-2 us)
        Tile t0, t1, t2;
        if (wave < ntile) load(t0, wave);
        if (wave + ST_NW < ntile) load(t1, wave + ST_NW);
        for (int rt = wave; rt < ntile; rt += 3 * ST_NW) {
            if (rt + 2 * ST_NW < ntile) load(t2, rt + 2 * ST_NW);
            finish(t0, rt);
            if (rt + ST_NW < ntile) {
                if (rt + 3 * ST_NW < ntile) load(t0, rt + 3 * ST_NW);
                finish(t1, rt + ST_NW);
            }
            if (rt + 2 * ST_NW < ntile) {
                if (rt + 4 * ST_NW < ntile) load(t1, rt + 4 * ST_NW);
                finish(t2, rt + 2 * ST_NW);
            }
        }
        if constexpr (NORM) if (d.part_out) { // (workgroup-uniform) fixed reduction order: lanes, then waves
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nrm2 += __shfl_xor(nrm2, o);
            __syncthreads(); // (W1s is free: phase 2 is over)
            if (lane == 0) W1s[0][wave] = nrm2;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < ST_NW; ++w) t += W1s[0][w];
                *(gp)d.part_out = t;
            }
        }
    }
    // ---- look-ahead inside the launch: the updated strip IS the next panel -- its 32-column latency chain runs while the
    //      other workgroups of this launch are still updating their strips (a stream-level look-ahead pays ~20 us per
    //      cross-stream dependency, this one nothing)
    // (a separate instantiation: with the panel body inlined the kernel needs scratch and 22 KB more LDS, which costs the
    //  plain strips of large matrices ~15 us per launch -- 41 -> 57 us at chi=4096 -- so they keep the lean kernel)
    if constexpr (LA) {
        if (d.next_off) { // workgroup-uniform
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads(); // every wave's stores have left for L2
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(la_flags + (d.next_off - 1), la_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if constexpr (FUSE_PANEL) {
        __shared__ PanelShared psh;
        if (d.next_off) { // workgroup-uniform
            __syncthreads(); // (the strip's stores are visible to the whole workgroup: same CU, write-through L1)
            const PanelDesc pd = *reinterpret_cast<const PanelDesc*>(reinterpret_cast<const char*>(descs) + d.next_off);
            panel_reg_body<RP_RPT_FUSED>(pd, psh, tid);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same update with the ROWS of a strip split over several workgroups, in two launches (round 3).  A strip workgroup of
// reflector_strip_kernel is a latency chain -- its eight waves stream 1442 rows in ~6 dependent batches of loads per phase,
// ~38 us whatever the rest of the chip does, and a panel step of a large matrix has only ~45 strips for 256 CUs.  Here a
// strip of mr rows becomes ceil(mr / chunk) workgroups of at most `chunk` rows (default 384):
//   strip_w1_kernel    : partial W1 = V_chunk^T C_chunk (phase 1 of the strip kernel on the chunk's rows) -> wpart[slot];
//   strip_apply_kernel : W1 = sum of the strip's partials (fixed order), W2 = P W1, C_chunk^T -= W2^T V_chunk^T (phases 2, 3).
// One batch of loads per phase instead of six; the launch boundary between the two replaces an exchange inside a launch.
struct ChunkDesc {
    double* C;       // the chunk's rows of the strip: element (i, c) at C[c*ldc + i], i < mr, c < nc
    const double* V; // the same rows of the reflector panel
    const double* T;
    double* wpart;   // partial W1's of the strip: n_slots x NBK*NBK doubles
    int64_t ldc, ldv;
    int32_t mr, nc, pw, transT;
    int32_t slot, n_slots, row0, step; // row0: first row of the chunk inside the strip (the rows below pw count for the stop test)
    const double* ctl;
    double* part_out;
};

__global__ void __launch_bounds__(ST_NT) strip_w1_kernel(const ChunkDesc* __restrict__ descs)
{
    __shared__ double part[ST_NW][NBK][ST_LS];
    const ChunkDesc d = descs[blockIdx.x];
    if (d.ctl) { // (workgroup-uniform)
        const double stopped = *(gcp)d.ctl;
        if (stopped != 0.0 && (double)d.step >= stopped - 1.0) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 15, kq = lane >> 4;
    gcp V = (gcp)d.V;
    gp C = (gp)d.C;
    const int mr = d.mr, nc = d.nc, pw = d.pw;
    const int64_t ldv = d.ldv, ldc = d.ldc;
    // ---- phase 1: partial W1 of this wave's row chunks (8 rows per step).  Columns beyond pw / nc are
    //      clamped to column 0: their products meet zero rows/columns of P or are never stored.
    {
        const int64_t vo0 = (int64_t)(x < pw ? x : 0) * ldv, vo1 = (int64_t)(x + 16 < pw ? x + 16 : 0) * ldv;
        const int64_t co0 = (int64_t)(x < nc ? x : 0) * ldc, co1 = (int64_t)(x + 16 < nc ? x + 16 : 0) * ldc;
        d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        const int nstep = (mr + 7) / 8;
        double va[ST_UN][2][2], ca[ST_UN][2][2], vn[ST_UN][2][2], cn[ST_UN][2][2]; // [step][tile][row of the pair]
        auto load = [&](double (&v)[ST_UN][2][2], double (&c)[ST_UN][2][2], int s0) {
#pragma unroll
            for (int u = 0; u < ST_UN; ++u) {
                const int i = (s0 + ST_NW * u) * 8 + 2 * kq;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool ok = i + h < mr;
                    const int ii = ok ? i + h : 0;
                    v[u][0][h] = ok ? V[vo0 + ii] : 0.0;
                    v[u][1][h] = ok ? V[vo1 + ii] : 0.0;
                    c[u][0][h] = ok ? C[co0 + ii] : 0.0;
                    c[u][1][h] = ok ? C[co1 + ii] : 0.0;
                }
            }
        };
        if (wave < nstep) load(va, ca, wave);
        for (int s0 = wave; s0 < nstep; s0 += ST_NW * ST_UN) {
            const bool more = s0 + ST_NW * ST_UN < nstep;
            if (more) load(vn, cn, s0 + ST_NW * ST_UN);
#pragma unroll
            for (int u = 0; u < ST_UN; ++u) {
                if (s0 + ST_NW * u < nstep) { // wave-uniform (rows beyond mr inside a step are loaded as zeros)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u][i][h], ca[u][j][h], acc[i][j], 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < ST_UN; ++u)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            va[u][i][h] = vn[u][i][h];
                            ca[u][i][h] = cn[u][i][h];
                        }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave][i * 16 + kq + 4 * r][j * 16 + x] = acc[i][j][r];
    }
    __syncthreads();
    gp out = (gp)d.wpart + (size_t)d.slot * (NBK * NBK);
    for (int e = tid; e < NBK * NBK; e += ST_NT) {
        const int a = e / NBK, c = e % NBK;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < ST_NW; ++w) t += part[w][a][c];
        out[e] = t;
    }
}

template <bool NORM>
__global__ void __launch_bounds__(ST_NT) strip_apply_kernel(const ChunkDesc* __restrict__ descs)
{
    __shared__ double W1s[NBK][ST_LS], W2s[NBK][ST_LS], Ps[NBK][ST_LS];
    const ChunkDesc d = descs[blockIdx.x];
    if (d.ctl) { // (workgroup-uniform)
        const double stopped = *(gcp)d.ctl;
        if (stopped != 0.0 && (double)d.step >= stopped - 1.0) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 15, kq = lane >> 4;
    gcp V = (gcp)d.V;
    gp C = (gp)d.C;
    gcp T = (gcp)d.T;
    const int mr = d.mr, nc = d.nc, pw = d.pw;
    const int64_t ldv = d.ldv, ldc = d.ldc;
    gcp wp = (gcp)d.wpart;
    for (int e = tid; e < NBK * NBK; e += ST_NT) {
        const int a = e / NBK, b = e % NBK;
        Ps[a][b] = T[d.transT ? b * NBK + a : e];
        double t = 0.0;
        for (int sl = 0; sl < d.n_slots; ++sl) t += wp[(size_t)sl * (NBK * NBK) + e]; // (the same order in every chunk of the strip)
        W1s[a][b] = t;
    }
    __syncthreads();
    // ---- phase 2: W2 = P W1, tile (wave >> 1, wave & 1)
    if (wave < 4) {
        const int ta = wave >> 1, tc = wave & 1;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NBK / 4; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ps[ta * 16 + x][4 * kk + kq], W1s[4 * kk + kq][tc * 16 + x], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) W2s[ta * 16 + kq + 4 * r][tc * 16 + x] = acc[r];
    }
    __syncthreads();
    // ---- phase 3: (C_s)^T tile [c][i] -= sum_a W2[a][c] V[i][a], 16 rows i per tile, tiles dealt to the waves
    {
        double w2[2][NBK / 4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kk = 0; kk < NBK / 4; ++kk) w2[ct][kk] = -W2s[4 * kk + kq][ct * 16 + x];
        int64_t vcol[NBK / 4], ccol[2][4];
#pragma unroll
        for (int kk = 0; kk < NBK / 4; ++kk) vcol[kk] = (int64_t)(4 * kk + kq < pw ? 4 * kk + kq : 0) * ldv;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = ct * 16 + kq + 4 * r;
                ccol[ct][r] = (int64_t)(c < nc ? c : 0) * ldc;
            }
        const int ntile = (mr + 15) / 16;
        struct Tile {
            double v[NBK / 4];
            d4 c[2];
        };
        auto load = [&](Tile& t, int rt) {
            const int i = rt * 16 + x;
            const bool ok = i < mr;
            const int ii = ok ? i : 0;
#pragma unroll
            for (int kk = 0; kk < NBK / 4; ++kk) t.v[kk] = ok ? V[vcol[kk] + ii] : 0.0;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) t.c[ct][r] = C[ccol[ct][r] + ii];
        };
        double nrm2 = 0.0; // squared norm of the entries this lane writes below the panel's rows
        auto finish = [&](Tile& t, int rt) {
#pragma unroll
            for (int kk = 0; kk < NBK / 4; ++kk) {
                t.c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[0][kk], t.v[kk], t.c[0], 0, 0, 0);
                t.c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[1][kk], t.v[kk], t.c[1], 0, 0, 0);
            }
            const int i = rt * 16 + x;
            if (i < mr) {
                const double below = i + d.row0 >= pw ? 1.0 : 0.0; // (row index inside the strip)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ct * 16 + kq + 4 * r < nc) {
                            C[ccol[ct][r] + i] = t.c[ct][r];
                            if constexpr (NORM) nrm2 = fma(below * t.c[ct][r], t.c[ct][r], nrm2);
                        }
            }
        };
        // three tiles in rotation: two are in flight while one is multiplied (a tile is ~0.4 us of MFMA, a miss ~1-2 us)
        Tile t0, t1, t2;
        if (wave < ntile) load(t0, wave);
        if (wave + ST_NW < ntile) load(t1, wave + ST_NW);
        for (int rt = wave; rt < ntile; rt += 3 * ST_NW) {
            if (rt + 2 * ST_NW < ntile) load(t2, rt + 2 * ST_NW);
            finish(t0, rt);
            if (rt + ST_NW < ntile) {
                if (rt + 3 * ST_NW < ntile) load(t0, rt + 3 * ST_NW);
                finish(t1, rt + ST_NW);
            }
            if (rt + 2 * ST_NW < ntile) {
                if (rt + 4 * ST_NW < ntile) load(t1, rt + 4 * ST_NW);
                finish(t2, rt + 2 * ST_NW);
            }
        }
        if constexpr (NORM) if (d.part_out) { // (workgroup-uniform) fixed reduction order: lanes, then waves
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nrm2 += __shfl_xor(nrm2, o);
            __syncthreads(); // (W1s is free: phase 2 is over)
            if (lane == 0) W1s[0][wave] = nrm2;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < ST_NW; ++w) t += W1s[0][w];
                *(gp)d.part_out = t;
            }
        }
    }
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
    hipLaunchKernelGGL(xpose_kernel, dim3(helper_grid_x(descs.size()), (unsigned)descs.size()), dim3(256), 0, ctx->stream, static_cast<const XposeDesc*>(d));
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

// The long-K product of a block-reflector application, W1 = V^T A (NBK x nt, K = remaining rows), has
// only nt/32 narrow tiles, each a serial walk over K: it is cut into up to kWSplit row chunks that
// run as independent problems, and the partials are summed for free as K-segments of the small
// product with T that follows (W2 = T^T sum_s W1_s = sum_s T^T W1_s).
constexpr int kWSplit = 8;
static inline int w_split(int64_t mr)
{
    static const int env = getenv("CYB_QR_WSPLIT") ? atoi(getenv("CYB_QR_WSPLIT")) : 0;
    if (env > 0) return (int)std::min<int64_t>(std::min(env, kWSplit), std::max<int64_t>(1, (mr + 255) / 256));
    // measured on the chi=4096 SVD list (1442 rows): 1 chunk 56.6 ms, 2: 53.1, 4: 53.6, 8: 55.1 -- every chunk
    // is one more K-segment of the rank-32 update that follows
    return mr >= 2600 ? 4 : mr >= 1800 ? 3 : mr >= 700 ? 2 : 1;
}

// The block-reflector application runs as ONE strip kernel per panel step (reflector_strip_kernel) unless
// CYB_QR_GEMM_UPDATE is set (the two-launch grouped-GEMM formulation, kept for comparison).
static inline bool use_strips()
{
    static const bool off = getenv("CYB_QR_GEMM_UPDATE") != nullptr;
    return !off;
}
static void add_strips(std::vector<StripDesc>& out, double* C, int64_t ldc, int64_t ncols, const double* V, int64_t ldv,
                       const double* T, int64_t mr, int pw, int transT, int64_t next_tag = -1)
{
    // next_tag >= 0: index of the PanelDesc (in the caller's list) that strip 0 factors after its update; resolved to a
    // device pointer once the image offsets are known
    for (int64_t c0 = 0; c0 < ncols; c0 += NBK)
        out.push_back(StripDesc{C + (size_t)c0 * ldc, V, T, ldc, ldv, (int32_t)mr, (int32_t)std::min<int64_t>(NBK, ncols - c0), pw, transT,
                                c0 == 0 && next_tag >= 0 ? next_tag + 1 : 0});
}
static void sort_strips(std::vector<StripDesc>& v)
{
    // longest strips first: the launch is one wave of workgroups, the tail is the longest strip
    std::stable_sort(v.begin(), v.end(), [](const StripDesc& a, const StripDesc& b) { return a.mr > b.mr; });
}

// ---- row-split strips (strip_w1_kernel + strip_apply_kernel)
// The chunk size is chosen PER PANEL STEP so that the chunk workgroups of the step fit the chip once (one workgroup per CU:
// the kernels use the whole register file): a single large matrix has ~45 strips per step and is cut into chunks of 384
// rows, a list whose strips fill the chip anyway is not cut at all (two launches of two rounds each measured SLOWER than the
// one-launch strip kernel there: chi=4096 list 26.3 -> 28.9 ms with a fixed 384, single 1442^2 block 20.9 -> 17.4 ms).
constexpr int kChunkRowsMin = 192;
static const int kChunkRowsChoices[] = {192, 256, 384, 512, 768, 1024, 1536};
static int strip_split_mode()
{
    // CYB_QR_CHUNK_ROWS: unset / -1 = adaptive, 0 = never split, n = fixed chunk rows (multiple of 16)
    static const int v = getenv("CYB_QR_CHUNK_ROWS") ? atoi(getenv("CYB_QR_CHUNK_ROWS")) : -1;
    return v < 0 ? -1 : v / 16 * 16;
}
int bqr_strip_slots(int64_t rows)
{
    if (strip_split_mode() == 0) return 1;
    const int ch = strip_split_mode() > 0 ? std::min(strip_split_mode(), kChunkRowsMin) : kChunkRowsMin; // (the bound for every choice)
    return (int)std::max<int64_t>(1, (rows + ch - 1) / ch);
}
// rows per chunk of a strip of mr rows cut into chunks of at most ch rows: even shares, multiples of 16 rows (ch == 0: one chunk)
static int strip_chunk_share(int mr, int ch)
{
    if (ch <= 0 || mr <= ch) return std::max(mr, 1);
    const int ns = (mr + ch - 1) / ch;
    return ((mr + ns - 1) / ns + 15) / 16 * 16;
}
static int strip_chunks(int mr, int ch)
{
    const int share = strip_chunk_share(mr, ch);
    return std::max(1, (mr + share - 1) / share);
}
// chunk rows of a panel step whose strips are (rows, count) pairs; 0: no split (the one-launch strip kernel)
static int choose_chunk_rows(const std::vector<std::pair<int, int>>& strips, int n_cu)
{
    const int mode = strip_split_mode();
    if (mode == 0) return 0;
    if (mode > 0) return mode;
    int64_t n_strips = 0;
    for (const auto& s : strips) n_strips += s.second;
    if (n_strips == 0 || n_strips * 4 > (int64_t)n_cu * 3) return 0; // the strips fill three quarters of the chip: leave them whole
    static const int ch_min = getenv("CYB_QR_CHUNK_MIN") ? atoi(getenv("CYB_QR_CHUNK_MIN")) : kChunkRowsMin;
    for (int ch : kChunkRowsChoices) {
        if (ch < ch_min) continue;
        int64_t tot = 0;
        bool any = false;
        for (const auto& s : strips) {
            const int c = strip_chunks(s.first, ch);
            tot += (int64_t)c * s.second;
            any = any || c > 1;
        }
        if (tot <= n_cu) return any ? ch : 0;
    }
    return 0;
}
// the chunk descriptors of a strip list; wbase: the partial-W1 workspace (NBK * NBK doubles per chunk)
static void make_chunks(const std::vector<StripDesc>& sd, double* wbase, int ch, std::vector<ChunkDesc>& out)
{
    out.clear();
    size_t slot_base = 0;
    for (const StripDesc& d : sd) {
        const int share = strip_chunk_share(d.mr, ch), nch = strip_chunks(d.mr, ch);
        for (int sl = 0; sl < nch; ++sl) {
            const int r0 = sl * share;
            out.push_back(ChunkDesc{d.C + r0, d.V + r0, d.T, wbase + slot_base * (size_t)(NBK * NBK), d.ldc, d.ldv, std::min(share, d.mr - r0), d.nc, d.pw,
                                    d.transT, sl, nch, r0, d.step, d.ctl, d.part_out ? d.part_out + sl : nullptr});
        }
        slot_base += (size_t)nch;
    }
}

size_t bqr_aux_bytes(int64_t m, int64_t n, int64_t ld, int64_t kc)
{
    const int64_t k = std::min(m, n);
    const int64_t npan = (k + NBK - 1) / NBK;
    return al256(sizeof(double) * (size_t)ld * (size_t)std::max<int64_t>(k, 1)) + al256(sizeof(double) * (size_t)npan * NBK * NBK) +
           al256(sizeof(double) * (size_t)std::max<int64_t>(k, 1)) + al256(sizeof(double) * (1 + kWSplit) * NBK * (size_t)std::max<int64_t>(std::max(n, kc), 1));
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
    off += al256(sizeof(double) * (1 + kWSplit) * NBK * (size_t)std::max<int64_t>(std::max<int64_t>(q.n, kc), 1));
    return off;
}

int bqr_factor(cyb_ctx_t ctx, const std::vector<BqrMat>& mats)
{
    auto pflags = [](const BqrMat& q) { return (q.v_zeroed ? PANEL_V_ZEROED : 0) | (q.reflect_always ? PANEL_REFLECT_ALWAYS : 0); };
    int max_pan = 0;
    for (const auto& q : mats) max_pan = std::max(max_pan, (q.k + NBK - 1) / NBK);
    // LOOK-AHEAD.  A panel step is [panel kernel: one workgroup per matrix, a latency chain of 32 column steps] ->
    // [trailing update: two grouped-GEMM launches over the whole chip].  The next panel only needs ITS 32 columns
    // updated, so the update is cut in two: the next panel's columns (two small launches, main stream) and the rest
    // (two launches on a side stream that leaves a few CUs free, context aux()).  The rest of step p then runs
    // BESIDE the panel kernel of step p + 1 instead of in front of it:
    //   main:  panel(p) . record P[p] . wait R[p-1] . next-columns(p) . panel(p+1) ...          (critical path)
    //   side:  wait P[p] . rest(p) . record R[p]
    // (columns >= j1 + 32 are written by rest(p) only, columns [j1, j1 + 32) by next-columns(p) and panel(p + 1) only;
    //  their scratch regions are disjoint).
    // MEASURED AND LEFT OFF (opt-in: CYB_QR_LOOKAHEAD=1): correct, but every panel step pays two cross-stream event
    // waits, and a dependency between two HIP streams costs more than the ~60 us of update it takes off the chain --
    // chi=4096 SVD list 50.2 -> 62.7 ms, one 1442 x 1442 block 37.0 -> 47.1 ms (profiles/README.md, round 2).
    static const bool want_la = getenv("CYB_QR_LOOKAHEAD") != nullptr;
    hipStream_t side = nullptr;
    const bool strips = use_strips();
    const bool lookahead = !strips && want_la && max_pan >= 6 && ctx->aux(&side) == CYB_OK;
    // Stage 1: the descriptors of EVERY panel step (they depend on shapes and pointers only) go into one host
    // image; stage 2: one upload; stage 3: the launches, panel step by panel step.
    struct Step {
        size_t off_pd = 0, off_pdr = 0;
        unsigned n_pd = 0, n_pdr = 0;
        GemmStaged s1, s3;     // whole trailing matrix (unsplit) or its next-panel columns (look-ahead)
        GemmStaged r1, r3;     // look-ahead: the rest of the trailing matrix
        bool has_rest = false;
        size_t off_sd = 0;     // strip formulation: the descriptors of this step's strips
        unsigned n_sd = 0;
        bool fused = false;    // some strip of this step factors the next panel (kernel instantiation with the panel body)
        bool norm = false;     // some strip of this step leaves its squared norm for the early-stop test (kernel instantiation with the accumulator)
        unsigned n_la = 0;     // role-split look-ahead: panel workgroups behind the strips (their descriptors at off_pn)
        size_t off_pn = 0;
        size_t off_cd = 0;     // row-split strips: chunk descriptors (strip_w1_kernel + strip_apply_kernel instead of the strip kernel)
        unsigned n_cd = 0;
        int wave = 0;          // every register-resident panel of this step is short enough for the one- / two- / four-wave kernel (1, 2, 4)
        size_t off_pdm = 0;    // panels of more than 1536 rows: several workgroups per matrix (qr_panel_multi_kernel)
        unsigned n_pdm = 0;
    };
    std::vector<Step> steps;
    std::vector<char> image;
    auto put = [&image](const void* src, size_t bytes) {
        const size_t off = (image.size() + 255) / 256 * 256;
        image.resize(off + bytes);
        memcpy(image.data() + off, src, bytes);
        return off;
    };
    // In-launch look-ahead (strip formulation): the workgroup that updates strip 0 of step p factors panel p + 1 right
    // after it (reflector_strip_kernel<true>) -- that panel's launch disappears and its latency chain runs beside the
    // other strips.  BUILT, CORRECT, AND LEFT OFF (opt-in: CYB_QR_FUSE=1; tests/test_gpu_decomp.py runs it): it buys nothing.
    // The kernel instantiation with the panel body needs scratch and 22 KB more LDS and is ~15 us slower per launch on
    // the strips of large matrices (41 -> 57 us at chi=4096: 37.6 -> 39.9 ms per step when it served every launch), so it
    // may only serve steps whose active matrices all have at most CYB_QR_FUSE_ROWS rows (default 768, two panel rows per
    // thread) -- and there, A/B on one box with the lean kernel back in place for everything else, the toy DMRG at
    // chi=256 runs 0.300 / 0.291 s per sweep with it and 0.303 / 0.287 without: a panel step of a 140-row matrix is the
    // 105 us latency chain of the panel either way, the ~8 us strip and one launch gap beside it do not show.
    static const bool no_fuse = getenv("CYB_QR_FUSE") == nullptr;
    static const int fuse_rows = getenv("CYB_QR_FUSE_ROWS") ? atoi(getenv("CYB_QR_FUSE_ROWS")) : 768;
    std::vector<char> fused(mats.size(), 0); // panel p of this matrix was factored inside step p - 1's strip launch
    // Role-split look-ahead (reflector_strip_kernel<.., LA>): panels of more than la_min_rows rows (the eight-wave register
    // kernel) are factored by an extra workgroup of the PREVIOUS step's strip launch.
    // BUILT, CORRECT (tests/test_gpu_decomp.py + test_gpu_fullsize.py green with it on), AND LEFT OFF (opt-in: CYB_QR_LA=1).
    // Round-3 A/B on one box, two alternations: one 1442 x 1442 block 24.66 / 24.75 ms with it, 24.68 / 24.96 without;
    // full-rank 1442 55.7-56.0 vs 55.6; chi=4096 15-block list 34.7 / 34.8 ms WITH vs 31.0 / 30.9 without; full-rank 3-block
    // list 65.6 / 65.9 vs 61.7 / 61.9.  Why it cannot win: every strip of a step takes the same ~38 us (each workgroup
    // streams the whole 1442 x 32 reflector panel), so the strip the panel waits for finishes WITH the others, not before
    // them -- the chain per step stays strip + panel (~40 + 117 us) and only the launch gap goes; and the instantiation with
    // the panel role (115 KB LDS, 12-28 B scratch) slows the strip role of multi-matrix launches by more than that.
    static const bool la_on = getenv("CYB_QR_LA") && atoi(getenv("CYB_QR_LA")) != 0;
    static const int la_min_rows = getenv("CYB_QR_LA_MIN") ? atoi(getenv("CYB_QR_LA_MIN")) : 4 * RP_WAVE_ROWS;
    unsigned int* la_flags = nullptr;
    const int la_err = (int)mats.size();
    bool la_used = false;
    // exchange buffers of the multi-workgroup panel kernel (matrices with more than 1536 rows), one region per matrix:
    // [tickets of all matrices | error word | per matrix: NBK x n_wg x 64 doubles]
    static const bool no_multi = getenv("CYB_QR_NOMULTI") != nullptr;
    std::vector<size_t> x_off(mats.size(), 0);
    size_t x_bytes = 0, n_multi_wg = 0;
    const size_t t_bytes = (sizeof(unsigned int) * NBK * mats.size() + 255) / 256 * 256;
    for (size_t qi = 0; qi < mats.size(); ++qi) {
        const int wmax = (mats[qi].m + RPM_ROWS - 1) / RPM_ROWS;
        if (mats[qi].m > RPM_ROWS && std::min(mats[qi].m, mats[qi].n) > 0) {
            x_off[qi] = t_bytes + 256 + x_bytes;
            x_bytes += sizeof(double) * NBK * (size_t)wmax * 64;
            n_multi_wg += (size_t)wmax;
        }
    }
    const bool multi = !no_multi && x_bytes > 0 && n_multi_wg <= (size_t)ctx->n_cu; // (all workgroups of a launch must be resident)
    char* xbase = nullptr;
    bool la_any = false; // some panel is tall enough for the role-split look-ahead: its flags live behind the exchange region
    if (la_on && use_strips())
        for (const auto& q : mats) la_any = la_any || (q.k > NBK && q.m - NBK > la_min_rows);
    if (multi || la_any) {
        const size_t x_total = multi ? (t_bytes + 256 + x_bytes + 255) / 256 * 256 : 0;
        void* xw = nullptr;
        CYB_TRY(ctx->workspace(x_total + 256 + sizeof(unsigned int) * (mats.size() + 1), &xw, 3));
        xbase = static_cast<char*>(xw);
        if (multi) CYB_HIP(hipMemsetAsync(xbase + t_bytes, 0, 256, ctx->stream)); // the error word
        if (la_any) {
            la_flags = reinterpret_cast<unsigned int*>(xbase + x_total);
            CYB_HIP(hipMemsetAsync(la_flags, 0, sizeof(unsigned int) * (mats.size() + 1), ctx->stream));
        }
    }
    // The descriptors are built, uploaded and launched in CHUNKS of panel steps (2, 6, 18, ... steps): the host builds the
    // next chunk while the device runs the previous ones.  With one image for the whole factorisation the device sat idle for
    // the 0.30-0.36 ms it takes to lay out 46 steps of a 15-matrix list (kernel trace of the chi=4096 step: the gap in front
    // of the first panel kernel of either QR).
    const bool multi_ok = multi; // (panels of more than 1536 rows stop too when they run on the multi-workgroup kernel)
    static const bool no_stop = getenv("CYB_QR_NOSTOP") != nullptr;
    static const bool no_reg_g = getenv("CYB_QR_PANEL_GLOBAL") != nullptr;
    auto stoppable = [&](const BqrMat& q) {
        return !no_stop && strips && !no_reg_g && q.ctl && q.parts && q.stop_rel2 > 0.0 && (q.m <= RP_NT * RP_RPT || multi_ok);
    };
    // row-split strips (two launches per step, see strip_w1_kernel): not together with the opt-in look-ahead forms
    const bool split = strips && strip_split_mode() != 0 && no_fuse && !la_on;
    double* wsplit = nullptr;
    if (split) {
        size_t n_ch = 0;
        for (const auto& q : mats) n_ch += (size_t)((q.n + NBK - 1) / NBK) * (size_t)bqr_strip_slots(q.m);
        void* w = nullptr;
        CYB_TRY(ctx->workspace(sizeof(double) * NBK * NBK * std::max<size_t>(n_ch, 1), &w, 4));
        wsplit = static_cast<double*>(w);
    }
    std::vector<int> n_parts_prev(mats.size(), 0); // strips the previous step's update wrote for this matrix
    static const int stop_every = std::max(1, getenv("CYB_QR_STOP_EVERY") ? atoi(getenv("CYB_QR_STOP_EVERY")) : 2);
    bool all_done = false;
    int last_rest = -1;
    const int side_cu = ctx->n_cu - (ctx->n_cu + 15) / 16; // persistent grid of the side stream: the CUs its mask leaves it
    int chunk = lookahead ? max_pan : 2;
    for (int c0 = 0; c0 < max_pan && !all_done; c0 += chunk, chunk *= 3) {
    const int c1 = std::min(max_pan, c0 + chunk);
    steps.clear();
    image.clear();
    for (int p = c0; p < c1; ++p) {
        std::vector<PanelDesc> pd, pd_reg, pd_next;
        std::vector<PanelDescM> pdm;
        std::vector<StripDesc> sd;
        GemmBatch g1, g3, h1, h3;
        int n_active = 0;
        int max_m = 0; // the fused kernel instantiation serves the whole launch: only when every active matrix is small
        for (const auto& q : mats)
            if (p * NBK < q.k) max_m = std::max(max_m, q.m);
        const bool step_fuse = !no_fuse && max_m <= fuse_rows;
        const bool step_la = la_on && strips && !step_fuse && la_flags != nullptr;
        bool step_has_la = false;
        bool step_norm = false;
        int step_ch = 0; // chunk rows of this step's strips (0: whole strips)
        if (split && !step_fuse) {
            std::vector<std::pair<int, int>> cnt;
            for (const auto& q : mats) {
                const int j0 = p * NBK;
                if (j0 >= q.k) continue;
                const int64_t nt = q.n - (j0 + std::min(NBK, q.k - j0));
                if (nt > 0) cnt.push_back({q.m - j0, (int)((nt + NBK - 1) / NBK)});
            }
            step_ch = choose_chunk_rows(cnt, ctx->n_cu);
        }
        for (size_t qi = 0; qi < mats.size(); ++qi) {
            const auto& q = mats[qi];
            const int j0 = p * NBK;
            if (j0 >= q.k) continue;
            ++n_active;
            const int pw = std::min(NBK, q.k - j0);
            static const bool no_reg = getenv("CYB_QR_PANEL_GLOBAL") != nullptr;
            if (fused[qi]) fused[qi] = 0;
            else if (!no_reg && q.m - j0 <= RP_NT * RP_RPT) {
                pd_reg.push_back(PanelDesc{q.Ac, q.V, q.T + (size_t)p * NBK * NBK, q.tau, q.ld, q.m, j0, pw, pflags(q)});
                if (stoppable(q)) {
                    PanelDesc& d = pd_reg.back();
                    d.ctl = q.ctl;
                    d.parts = q.parts;
                    d.rel2 = q.stop_rel2;
                    d.n_parts = n_parts_prev[qi];
                    d.step = p;
                }
            } else if (multi && !no_reg) {
                const int n_wg = (q.m - j0 + RPM_ROWS - 1) / RPM_ROWS;
                for (int w = 0; w < n_wg; ++w)
                {
                    pdm.push_back(PanelDescM{q.Ac, q.V, q.T + (size_t)p * NBK * NBK, q.tau, reinterpret_cast<double*>(xbase + x_off[qi]),
                                             reinterpret_cast<unsigned int*>(xbase) + qi * NBK, reinterpret_cast<unsigned int*>(xbase + t_bytes),
                                             q.ld, q.m, j0, pw, pflags(q), w, n_wg, 0, 0});
                    if (stoppable(q)) {
                        PanelDescM& d = pdm.back();
                        d.ctl = q.ctl;
                        d.parts = q.parts;
                        d.rel2 = q.stop_rel2;
                        d.n_parts = n_parts_prev[qi];
                        d.step = p;
                    }
                }
            } else
                pd.push_back(PanelDesc{q.Ac, q.V, q.T + (size_t)p * NBK * NBK, q.tau, q.ld, q.m, j0, pw, pflags(q) & PANEL_REFLECT_ALWAYS});
            const int j1 = j0 + pw;
            const int64_t nt = q.n - j1, mr = q.m - j0;
            const int parts_lag = n_parts_prev[qi];
            n_parts_prev[qi] = 0;
            if (nt <= 0) continue;
            double* W1 = q.scratch + q.scr_half; // up to kWSplit partials of scr_half doubles
            const double* Vp = q.V + (size_t)j0 * q.ld + j0;        // (i,a) at a*ld + i
            const double* Tp = q.T + (size_t)p * NBK * NBK;
            if (strips) {
                int64_t tag = -1;
                if (step_fuse && !no_reg && !stoppable(q) && j1 < q.k && q.m - j1 <= RP_NT * RP_RPT_FUSED) {
                    tag = (int64_t)pd_next.size();
                    pd_next.push_back(PanelDesc{q.Ac, q.V, q.T + (size_t)(p + 1) * NBK * NBK, q.tau, q.ld, q.m, j1, std::min(NBK, q.k - j1), pflags(q)});
                    fused[qi] = 1;
                } else if (step_la && !no_reg && j1 < q.k && q.m - j1 <= RP_NT * RP_RPT && q.m - j1 > la_min_rows) {
                    tag = (int64_t)pd_next.size();
                    pd_next.push_back(PanelDesc{q.Ac, q.V, q.T + (size_t)(p + 1) * NBK * NBK, q.tau, q.ld, q.m, j1, std::min(NBK, q.k - j1), pflags(q)});
                    if (stoppable(q)) {
                        // the panel runs BESIDE this step's strips: it may only read the norms the previous step left, and only
                        // if this step does not overwrite them (the stop then comes one panel later than without look-ahead)
                        const bool measure = p % stop_every == stop_every - 1;
                        PanelDesc& d = pd_next.back();
                        d.ctl = q.ctl;
                        d.parts = q.parts;
                        d.rel2 = q.stop_rel2;
                        d.n_parts = measure ? 0 : parts_lag;
                        d.step = p + 1;
                    }
                    fused[qi] = 1;
                    step_has_la = true;
                }
                const size_t s_begin = sd.size();
                add_strips(sd, q.Ac + (size_t)j1 * q.ld + j0, q.ld, nt, Vp, q.ld, Tp, mr, pw, 1, tag);
                if (stoppable(q)) {
                    const bool measure = p % stop_every == stop_every - 1; // (the trailing norm is measured every stop_every-th step)
                    const int nch = strip_chunks((int)mr, step_ch); // (every strip of a matrix has the same rows: the same chunks)
                    for (size_t si = s_begin; si < sd.size(); ++si) {
                        sd[si].ctl = q.ctl;
                        sd[si].part_out = measure ? q.parts + (si - s_begin) * (size_t)nch : nullptr;
                        sd[si].step = p;
                    }
                    if (measure) {
                        n_parts_prev[qi] = (int)(sd.size() - s_begin) * nch;
                        step_norm = true;
                    }
                }
                continue;
            }
            // W2_s (pw x nt) = T^T (Vp[rows of chunk s]^T At[rows of chunk s]): the T factor is applied in the
            // epilogue of the product (left factor of the grouped GEMM), the row-chunk partials are summed
            // as K-segments of the rank-pw update:  At^T (nt x mr, ld) -= sum_s W2_s^T Vp^T
            const int ns = w_split(mr);
            const int64_t chunk = ((mr + ns - 1) / ns + 15) / 16 * 16;
            // column ranges of the trailing matrix: [0, nn) = the next panel (or everything), [nn, nt) = the rest
            const int64_t nn = lookahead ? std::min<int64_t>(NBK, nt) : nt;
            auto add = [&](GemmBatch& a1, GemmBatch& a3, int64_t c0, int64_t nc, size_t scr_off) {
                if (nc <= 0) return;
                double* At = q.Ac + (size_t)(j1 + c0) * q.ld + j0;  // (i,c) at c*ld + i
                const int32_t seg0 = (int32_t)a3.segs.size();
                for (int sidx = 0; sidx < ns; ++sidx) {
                    const int64_t r0 = chunk * sidx, rows = std::min(chunk, mr - r0);
                    if (rows <= 0) break;
                    double* W2s = W1 + (size_t)sidx * q.scr_half + scr_off;
                    a1.add_post(W2s, pw, nc, nc, Vp + r0, q.ld, 1, At + r0, 1, q.ld, rows, 1.0, 0.0, Tp, 1, NBK);
                    a3.segs.push_back(cyb_gemm_seg{W2s, Vp, pw, 1, nc, q.ld, 1});
                }
                a3.probs.push_back(cyb_gemm_prob{At, nc, mr, q.ld, seg0, (int32_t)a3.segs.size(), -1.0, 1.0});
            };
            add(g1, g3, 0, nn, 0);
            add(h1, h3, nn, nt - nn, (size_t)NBK * NBK); // (scratch: NBK*NBK doubles for the next-panel part, the rest behind it)
        }
        if (n_active == 0) {
            all_done = true;
            break;
        }
        Step st;
        st.n_pd = (unsigned)pd.size();
        st.n_pdr = (unsigned)pd_reg.size();
        if (st.n_pd) st.off_pd = put(pd.data(), sizeof(PanelDesc) * pd.size());
        if (st.n_pdr) st.off_pdr = put(pd_reg.data(), sizeof(PanelDesc) * pd_reg.size());
        st.n_pdm = (unsigned)pdm.size();
        if (st.n_pdm) st.off_pdm = put(pdm.data(), sizeof(PanelDescM) * pdm.size());
        static const bool no_wave = getenv("CYB_QR_NOWAVE") != nullptr;
        int max_rows = 0;
        for (const auto& d : pd_reg) max_rows = std::max(max_rows, d.m - d.j0);
        st.wave = (no_wave || st.n_pdr == 0) ? 0 : max_rows <= RP_WAVE_ROWS ? 1 : max_rows <= 2 * RP_WAVE_ROWS ? 2 : max_rows <= 4 * RP_WAVE_ROWS ? 4 : 0;
        if (!sd.empty()) {
            sort_strips(sd);
            st.n_sd = (unsigned)sd.size();
            st.fused = !pd_next.empty() && !step_has_la;
            st.norm = step_norm;
            const size_t off_pn = pd_next.empty() ? 0 : put(pd_next.data(), sizeof(PanelDesc) * pd_next.size());
            const size_t off_sd = (image.size() + 255) / 256 * 256; // (where put() will place the strips)
            if (step_has_la) { // (next_off stays the index + 1 of the panel entry = of its flag)
                st.n_la = (unsigned)pd_next.size();
                st.off_pn = off_pn;
                la_used = true;
            } else
                for (auto& d : sd)
                    if (d.next_off) d.next_off = (int64_t)(off_pn + (size_t)(d.next_off - 1) * sizeof(PanelDesc)) - (int64_t)off_sd;
            st.off_sd = put(sd.data(), sizeof(StripDesc) * sd.size());
            if (step_ch > 0 && !st.fused && !st.n_la) {
                std::vector<ChunkDesc> cd;
                make_chunks(sd, wsplit, step_ch, cd);
                st.n_cd = (unsigned)cd.size();
                st.off_cd = put(cd.data(), sizeof(ChunkDesc) * cd.size());
            }
        }
        if (!g1.empty()) {
            CYB_TRY(g1.stage(ctx, image, st.s1));
            CYB_TRY(g3.stage(ctx, image, st.s3));
        }
        if (!h1.empty()) {
            CYB_TRY(h1.stage(ctx, image, st.r1));
            CYB_TRY(h3.stage(ctx, image, st.r3));
            st.has_rest = true;
        }
        steps.push_back(st);
    }
    if (steps.empty()) break;
    void* d_image = nullptr;
    CYB_TRY(ctx->upload(image.data(), image.size(), &d_image));
    char* dbase = static_cast<char*>(d_image);
    if (lookahead) CYB_TRY(ctx->events(2 * steps.size()));
    for (size_t p = 0; p < steps.size(); ++p) {
        const Step& st = steps[p];
        if (st.n_pd)
            hipLaunchKernelGGL(qr_panel_kernel, dim3(st.n_pd), dim3(PNT), 0, ctx->stream,
                               reinterpret_cast<const PanelDesc*>(dbase + st.off_pd));
        if (st.n_pdm) {
            CYB_HIP(hipMemsetAsync(xbase, 0, t_bytes, ctx->stream)); // the tickets of every matrix
            hipLaunchKernelGGL(qr_panel_multi_kernel, dim3(st.n_pdm), dim3(RP_NT), 0, ctx->stream,
                               reinterpret_cast<const PanelDescM*>(dbase + st.off_pdm));
        }
        if (st.n_pdr && st.wave == 1)
            hipLaunchKernelGGL(qr_panel_wave_kernel, dim3(st.n_pdr), dim3(64), 0, ctx->stream,
                               reinterpret_cast<const PanelDesc*>(dbase + st.off_pdr));
        else if (st.n_pdr && st.wave == 2)
            hipLaunchKernelGGL(qr_panel_wave2_kernel, dim3(st.n_pdr), dim3(128), 0, ctx->stream,
                               reinterpret_cast<const PanelDesc*>(dbase + st.off_pdr));
        else if (st.n_pdr && st.wave == 4)
            hipLaunchKernelGGL(qr_panel_wave4_kernel, dim3(st.n_pdr), dim3(256), 0, ctx->stream,
                               reinterpret_cast<const PanelDesc*>(dbase + st.off_pdr));
        else if (st.n_pdr)
            hipLaunchKernelGGL(qr_panel_reg_kernel, dim3(st.n_pdr), dim3(RP_NT), 0, ctx->stream,
                               reinterpret_cast<const PanelDesc*>(dbase + st.off_pdr));
        CYB_HIP(hipGetLastError());
        if (lookahead && st.has_rest) {
            hipEvent_t evP = ctx->ev_pool[2 * p], evR = ctx->ev_pool[2 * p + 1];
            CYB_HIP(hipEventRecord(evP, ctx->stream));
            CYB_HIP(hipStreamWaitEvent(side, evP, 0));
            CYB_TRY(gemm_launch_staged(ctx, st.r1, d_image, side, side_cu));
            CYB_TRY(gemm_launch_staged(ctx, st.r3, d_image, side, side_cu));
            CYB_HIP(hipEventRecord(evR, side));
            if (last_rest >= 0) CYB_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_pool[2 * (size_t)last_rest + 1], 0));
            last_rest = (int)p;
        } else if (lookahead && last_rest >= 0) {
            CYB_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_pool[2 * (size_t)last_rest + 1], 0));
            last_rest = -1;
        }
        if (st.n_sd) {
            const StripDesc* sdp = reinterpret_cast<const StripDesc*>(dbase + st.off_sd);
            const PanelDesc* pnp = reinterpret_cast<const PanelDesc*>(dbase + st.off_pn);
            const unsigned la_tag = (unsigned)(c0 + (int)p + 1); // (unique per panel step of this call; the flags start at zero)
            const PanelDesc* no_pn = nullptr;
            unsigned int* no_fl = nullptr;
            if (st.n_cd) {
                const ChunkDesc* cdp = reinterpret_cast<const ChunkDesc*>(dbase + st.off_cd);
                hipLaunchKernelGGL(strip_w1_kernel, dim3(st.n_cd), dim3(ST_NT), 0, ctx->stream, cdp);
                if (st.norm) hipLaunchKernelGGL(strip_apply_kernel<true>, dim3(st.n_cd), dim3(ST_NT), 0, ctx->stream, cdp);
                else hipLaunchKernelGGL(strip_apply_kernel<false>, dim3(st.n_cd), dim3(ST_NT), 0, ctx->stream, cdp);
            } else if (st.n_la && st.norm)
                hipLaunchKernelGGL((reflector_strip_kernel<false, true, true>), dim3(st.n_sd + st.n_la), dim3(ST_NT), 0, ctx->stream, sdp,
                                   (int)st.n_sd, pnp, la_flags, la_tag, la_err);
            else if (st.n_la)
                hipLaunchKernelGGL((reflector_strip_kernel<false, false, true>), dim3(st.n_sd + st.n_la), dim3(ST_NT), 0, ctx->stream, sdp,
                                   (int)st.n_sd, pnp, la_flags, la_tag, la_err);
            else if (st.fused)
                hipLaunchKernelGGL(reflector_strip_kernel<true>, dim3(st.n_sd), dim3(ST_NT), 0, ctx->stream, sdp, (int)st.n_sd, no_pn, no_fl, 0u, 0);
            else if (st.norm)
                hipLaunchKernelGGL((reflector_strip_kernel<false, true>), dim3(st.n_sd), dim3(ST_NT), 0, ctx->stream, sdp, (int)st.n_sd, no_pn, no_fl, 0u, 0);
            else
                hipLaunchKernelGGL(reflector_strip_kernel<false>, dim3(st.n_sd), dim3(ST_NT), 0, ctx->stream, sdp, (int)st.n_sd, no_pn, no_fl, 0u, 0);
            CYB_HIP(hipGetLastError());
        }
        CYB_TRY(gemm_launch_staged(ctx, st.s1, d_image));
        CYB_TRY(gemm_launch_staged(ctx, st.s3, d_image));
    }
    } // chunks of panel steps
    if (last_rest >= 0) CYB_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_pool[2 * (size_t)last_rest + 1], 0)); // join
    if (la_used) { // (a wait that ran out of time means wrong factors, not a hang)
        unsigned int h_err = 0;
        CYB_HIP(hipMemcpyAsync(&h_err, la_flags + la_err, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        CYB_HIP(hipStreamSynchronize(ctx->stream));
        if (h_err) {
            set_error("blocked QR: a look-ahead panel workgroup waited more than a second for its strip");
            return CYB_ERR_HIP;
        }
    }
    if (multi) { // (the rare path pays one read-back: a poll that ran out of time means wrong factors, not a hang)
        unsigned int h_err = 0;
        CYB_HIP(hipMemcpyAsync(&h_err, xbase + t_bytes, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        CYB_HIP(hipStreamSynchronize(ctx->stream));
        if (h_err) {
            set_error("blocked QR: a workgroup of the multi-workgroup panel kernel waited more than a second for its partners");
            return CYB_ERR_HIP;
        }
    }
    return CYB_OK;
}

int bqr_apply_q(cyb_ctx_t ctx, const std::vector<BqrMat>& mats, const std::vector<BqrTarget>& targets)
{
    int max_pan = 0;
    for (const auto& t : targets) max_pan = std::max(max_pan, (mats[(size_t)t.mat].k + NBK - 1) / NBK);
    const bool strips = use_strips();
    struct Step {
        GemmStaged s1, s3;
        size_t off_sd = 0;
        unsigned n_sd = 0;
        size_t off_cd = 0; // row-split strips (see bqr_factor)
        unsigned n_cd = 0;
    };
    const bool split = strips && strip_split_mode() != 0;
    double* wsplit = nullptr;
    if (split) {
        size_t n_ch = 0;
        for (const auto& t : targets) n_ch += (size_t)((t.kc + NBK - 1) / NBK) * (size_t)bqr_strip_slots(mats[(size_t)t.mat].m);
        void* w = nullptr;
        CYB_TRY(ctx->workspace(sizeof(double) * NBK * NBK * std::max<size_t>(n_ch, 1), &w, 4));
        wsplit = static_cast<double*>(w);
    }
    std::vector<Step> steps; // panel steps staged in chunks (2, 6, 18, ... steps), one descriptor upload per chunk: the host
    std::vector<char> image; // lays out the next chunk while the device runs the previous ones (see bqr_factor)
    int chunk = 2;
    for (int c0 = max_pan - 1; c0 >= 0; c0 -= chunk, chunk *= 3) {
    const int c1 = std::max(-1, c0 - chunk); // steps c0, c0 - 1, ..., c1 + 1
    steps.clear();
    image.clear();
    for (int p = c0; p > c1; --p) {
        GemmBatch g1, g3;
        std::vector<StripDesc> sd;
        for (const auto& t : targets) {
            const BqrMat& q = mats[(size_t)t.mat];
            const int j0 = p * NBK;
            if (j0 >= q.k || t.kc <= 0) continue;
            const int pw = std::min(NBK, q.k - j0);
            const int64_t mr = q.m - j0;
            double* W1 = q.scratch + q.scr_half;
            CYB_REQUIRE((int64_t)NBK * t.kc <= q.scr_half, "bqr_apply_q: target wider than the carved scratch");
            const double* Vp = q.V + (size_t)j0 * q.ld + j0;
            double* Ct = t.C + j0; // rows j0.. of every column
            const double* Tp = q.T + (size_t)p * NBK * NBK;
            if (strips) {
                const size_t s_begin = sd.size();
                add_strips(sd, Ct, t.ldc, t.kc, Vp, q.ld, Tp, mr, pw, 0);
                if (q.ctl && q.stop_rel2 > 0.0) // (panels the factorisation skipped are the identity: their strips return at once)
                    for (size_t si = s_begin; si < sd.size(); ++si) {
                        sd[si].ctl = q.ctl;
                        sd[si].step = p;
                    }
                continue;
            }
            // W2_s = T (Vp[chunk s]^T C[chunk s]) (T in the epilogue);  C^T -= sum_s W2_s^T Vp^T
            const int ns = w_split(mr);
            const int64_t chunk = ((mr + ns - 1) / ns + 15) / 16 * 16;
            const int32_t seg0 = (int32_t)g3.segs.size();
            for (int sidx = 0; sidx < ns; ++sidx) {
                const int64_t r0 = chunk * sidx, rows = std::min(chunk, mr - r0);
                if (rows <= 0) break;
                double* W2s = W1 + (size_t)sidx * q.scr_half;
                g1.add_post(W2s, pw, t.kc, t.kc, Vp + r0, q.ld, 1, Ct + r0, 1, t.ldc, rows, 1.0, 0.0, Tp, NBK, 1);
                g3.segs.push_back(cyb_gemm_seg{W2s, Vp, pw, 1, t.kc, q.ld, 1});
            }
            g3.probs.push_back(cyb_gemm_prob{Ct, t.kc, mr, t.ldc, seg0, (int32_t)g3.segs.size(), -1.0, 1.0});
        }
        if (g1.empty() && sd.empty()) continue;
        Step st;
        if (!g1.empty()) {
            CYB_TRY(g1.stage(ctx, image, st.s1));
            CYB_TRY(g3.stage(ctx, image, st.s3));
        }
        if (!sd.empty()) {
            sort_strips(sd);
            st.n_sd = (unsigned)sd.size();
            const size_t off = (image.size() + 255) / 256 * 256;
            image.resize(off + sizeof(StripDesc) * sd.size());
            memcpy(image.data() + off, sd.data(), sizeof(StripDesc) * sd.size());
            st.off_sd = off;
            std::vector<std::pair<int, int>> cnt;
            for (const auto& d : sd) cnt.push_back({d.mr, 1});
            const int step_ch = split ? choose_chunk_rows(cnt, ctx->n_cu) : 0;
            if (step_ch > 0) {
                std::vector<ChunkDesc> cd;
                make_chunks(sd, wsplit, step_ch, cd);
                st.n_cd = (unsigned)cd.size();
                const size_t offc = (image.size() + 255) / 256 * 256;
                image.resize(offc + sizeof(ChunkDesc) * cd.size());
                memcpy(image.data() + offc, cd.data(), sizeof(ChunkDesc) * cd.size());
                st.off_cd = offc;
            }
        }
        steps.push_back(st);
    }
    if (steps.empty()) continue;
    void* d_image = nullptr;
    CYB_TRY(ctx->upload(image.data(), image.size(), &d_image));
    for (const auto& st : steps) {
        if (st.n_cd) {
            const ChunkDesc* cdp = reinterpret_cast<const ChunkDesc*>(static_cast<char*>(d_image) + st.off_cd);
            hipLaunchKernelGGL(strip_w1_kernel, dim3(st.n_cd), dim3(ST_NT), 0, ctx->stream, cdp);
            hipLaunchKernelGGL(strip_apply_kernel<false>, dim3(st.n_cd), dim3(ST_NT), 0, ctx->stream, cdp);
            CYB_HIP(hipGetLastError());
        } else if (st.n_sd) {
            hipLaunchKernelGGL(reflector_strip_kernel<false>, dim3(st.n_sd), dim3(ST_NT), 0, ctx->stream,
                               reinterpret_cast<const StripDesc*>(static_cast<char*>(d_image) + st.off_sd), (int)st.n_sd,
                               static_cast<const PanelDesc*>(nullptr), static_cast<unsigned int*>(nullptr), 0u, 0);
            CYB_HIP(hipGetLastError());
        }
        CYB_TRY(gemm_launch_staged(ctx, st.s1, d_image));
        CYB_TRY(gemm_launch_staged(ctx, st.s3, d_image));
    }
    } // chunks of panel steps
    return CYB_OK;
}

} // namespace cyb
