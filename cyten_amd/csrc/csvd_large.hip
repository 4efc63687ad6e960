// complex128 SVD / Hermitian eigh of blocks beyond the in-LDS limit of csvd_small.hip: complex one-sided Jacobi with the
// work matrix in device memory, one launch per round of the round-robin tournament for ALL blocks of the list.
//
// Same mathematics as csvd_small.hip (reference semantics: NumpyBlockBackend::matrix_svd numpy.cpp:1247-1297, ::eigh
// :658-680 on complex128 blocks): W = A V with the columns of W pairwise orthogonalised by the 2x2 unitaries
// [phase e^{-i arg g}, then the real Jacobi rotation]; a wide block is worked on as its conjugate transpose; eigh is the
// SVD-mode run on H + ||H||_F I whose right vectors are the eigenvectors.  What is different from the small kernel:
//   * a round is a kernel launch (one workgroup per column pair, columns streamed from L2 / HBM), a sweep ends with one
//     read-back of the per-block convergence measure;
//   * rank-deficient blocks (every theta = A B is one) get their missing left vectors by BLOCK completion: pseudo-random
//     candidate columns are projected against the finished vectors twice (two tiled complex products per pass), made
//     orthogonal among themselves by the same Jacobi rounds (no accumulation) and normalised.
// This is the functional path for large complex blocks: plain FMA arithmetic, no MFMA blocking (DESIGN.md section 4.5b
// gives its measured times next to the real pipeline's); the real dtype never comes here.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace cyb_clarge {

struct Req { // one block, as the C-ABI entries of csvd_small.hip hand it over
    const double* A;
    double *U, *S, *Vh; // eigh: U = eigenvectors, S = eigenvalues (ascending), Vh unused
    int64_t lda, ldu, ldvh;
    int32_t m, n, mode;
};
int run(cyb_ctx_t ctx, const std::vector<Req>& req, int32_t* sweeps_out);


} // namespace cyb_clarge

namespace {

constexpr int NT = 256;
constexpr int MAX_SWEEPS = 80;

typedef double d2 __attribute__((ext_vector_type(2))); // (re, im)

__device__ __forceinline__ d2 cmul(d2 a, d2 b) { return d2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ d2 cmulc(d2 a, d2 b) { return d2{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x}; } // conj(a) * b

struct Blk { // device image of one block
    const d2* A;
    d2 *U, *Vh;
    double* S;
    int64_t lda, ldu, ldvh;
    d2 *W, *V;      // W: Np columns of length M (column c at W + c * M); V: Np x Np, column c at V + c * Np
    double* stat;   // [0] scale (power of two), [1] ||scaled A||_F^2 (+ shift part), [2] shift, [3] null2
    double *sig, *cn;
    int32_t *rank, *kept, *nul, *cnt; // cnt[0] = number of null columns, cnt[1] = number of kept columns
    d2* P;
    int32_t m, n, mode, tall, M, N, Np, pad;
};

struct Jac { // one Jacobi problem of a round launch
    d2 *W, *V;           // V == nullptr: no accumulation
    const int32_t* cols; // nullptr: columns 0 .. ncol-1; else the physical column of each virtual one (-1: absent)
    uint32_t* off;       // max |g| / (|p| |q|) of the running sweep, float bits
    const double* stat;  // null2 at stat[3] (ignored when V == nullptr: candidates have full rank)
    int32_t M, Np, ncol, pair_base, active, pad;
};

__device__ double wg_sum(double v, double* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}
__device__ double wg_max(double v, double* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s = fmax(s, red[w]);
    return s;
}

// ---- scale / norm of the input, one workgroup per block (two passes over A)
__global__ void __launch_bounds__(1024) cl_prep_kernel(const Blk* __restrict__ blks)
{
    __shared__ double red[16];
    const Blk d = blks[blockIdx.x];
    const int64_t total = (int64_t)d.m * d.n;
    double amax = 0.0;
    for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
        const d2 v = d.A[(e / d.n) * d.lda + (e % d.n)];
        amax = fmax(amax, fmax(fabs(v.x), fabs(v.y)));
    }
    amax = wg_max(amax, red);
    const double scl = (amax > 0.0 && amax < __builtin_huge_val()) ? scalbn(1.0, -ilogb(amax)) : 1.0;
    double fro2 = 0.0;
    for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
        const d2 v = d.A[(e / d.n) * d.lda + (e % d.n)] * scl;
        fro2 += v.x * v.x + v.y * v.y;
    }
    fro2 = wg_sum(fro2, red);
    if (threadIdx.x == 0) {
        double shift = 0.0;
        if (d.mode == 1) {
            shift = sqrt(fro2);
            fro2 += shift * shift * d.N;
        }
        d.stat[0] = scl;
        d.stat[1] = fro2;
        d.stat[2] = shift;
        d.stat[3] = fro2 * (2.3e-16 * d.M) * (2.3e-16 * d.M);
    }
}

// ---- W = scaled A (or its conjugate transpose) [+ shift on the diagonal], V = identity
__global__ void __launch_bounds__(NT) cl_load_kernel(const Blk* __restrict__ blks)
{
    const Blk d = blks[blockIdx.y];
    const double scl = d.stat[0], shift = d.stat[2];
    const int64_t nw = (int64_t)d.Np * d.M, nv = d.V ? (int64_t)d.Np * d.Np : 0;
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < nw + nv; e += stride) {
        if (e < nw) {
            const int64_t c = e / d.M, r = e - c * d.M;
            d2 v = d2{0.0, 0.0};
            if (c < d.N) {
                if (d.tall) v = d.A[r * d.lda + c];
                else {
                    v = d.A[c * d.lda + r];
                    v.y = -v.y;
                }
                v *= scl;
                if (d.mode == 1 && r == c) v.x += shift;
            }
            d.W[e] = v;
        } else {
            const int64_t f = e - nw;
            const int64_t c = f / d.Np, r = f - c * d.Np;
            d.V[f] = d2{(c == r) ? 1.0 : 0.0, 0.0};
        }
    }
}

// ---- one round of the tournament: workgroup = one column pair of one problem
__global__ void __launch_bounds__(NT) cl_round_kernel(const Jac* __restrict__ probs, int nprob, int round)
{
    __shared__ double red[4][NT / 64];
    int b = 0;
    while (b + 1 < nprob && (int)blockIdx.x >= probs[b + 1].pair_base) ++b;
    const Jac d = probs[b];
    const int k = (int)blockIdx.x - d.pair_base;
    const int npairs = d.ncol / 2;
    if (!d.active || k < 0 || k >= npairs || round >= d.ncol - 1) return;
    int p = (k == 0) ? d.ncol - 1 : (round + k) % (d.ncol - 1);
    int q = (k == 0) ? round : (round - k + d.ncol - 1) % (d.ncol - 1);
    if (d.cols) {
        p = d.cols[p];
        q = d.cols[q];
        if (p < 0 || q < 0) return;
    }
    const int tid = threadIdx.x;
    d2* wp = d.W + (int64_t)p * d.M;
    d2* wq = d.W + (int64_t)q * d.M;
    double app = 0.0, aqq = 0.0, gr = 0.0, gi = 0.0;
    for (int i = tid; i < d.M; i += NT) {
        const d2 x = wp[i], y = wq[i];
        app += x.x * x.x + x.y * x.y;
        aqq += y.x * y.x + y.y * y.y;
        gr += x.x * y.x + x.y * y.y; // conj(x) * y
        gi += x.x * y.y - x.y * y.x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        app += __shfl_xor(app, o);
        aqq += __shfl_xor(aqq, o);
        gr += __shfl_xor(gr, o);
        gi += __shfl_xor(gi, o);
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = app;
        red[1][tid >> 6] = aqq;
        red[2][tid >> 6] = gr;
        red[3][tid >> 6] = gi;
    }
    __syncthreads();
    app = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    aqq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    gr = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    gi = red[3][0] + red[3][1] + red[3][2] + red[3][3];
    const double null2 = d.V ? d.stat[3] : 0.0;
    const double ag = sqrt(gr * gr + gi * gi);
    const double den = sqrt(app) * sqrt(aqq);
    if (!(app > null2 && aqq > null2 && ag > 1e-15 * den)) return;
    if (tid == 0) atomicMax(d.off, __float_as_uint((float)fmin(ag / den, 1.0)));
    const d2 ph = d2{gr / ag, -gi / ag}; // e^{-i arg g}
    const double zeta = (aqq - app) / (2.0 * ag);
    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
    for (int i = tid; i < d.M; i += NT) {
        const d2 x = wp[i], y = cmul(wq[i], ph);
        wp[i] = c * x - s * y;
        wq[i] = s * x + c * y;
    }
    if (d.V) {
        d2* vp = d.V + (int64_t)p * d.Np;
        d2* vq = d.V + (int64_t)q * d.Np;
        for (int i = tid; i < d.Np; i += NT) {
            const d2 x = vp[i], y = cmul(vq[i], ph);
            vp[i] = c * x - s * y;
            vq[i] = s * x + c * y;
        }
    }
}

// ---- column norms: out[c] = |W[:, c]| for the columns of `list` (nullptr: all Np columns; the padding column gets -1)
__global__ void __launch_bounds__(NT) cl_norms_kernel(const Blk* __restrict__ blks, int which)
{
    __shared__ double red[NT / 64];
    const Blk d = blks[blockIdx.y];
    int c = blockIdx.x;
    double* out = d.sig;
    if (which == 1) { // the completed (null) columns
        if (c >= d.cnt[0]) return;
        c = d.nul[c];
        out = d.cn;
    } else if (c >= d.Np) return;
    const d2* w = d.W + (int64_t)c * d.M;
    double s2 = 0.0;
    for (int i = threadIdx.x; i < d.M; i += NT) {
        const d2 v = w[i];
        s2 += v.x * v.x + v.y * v.y;
    }
    s2 = wg_sum(s2, red);
    if (threadIdx.x == 0) out[c] = (c < d.N) ? sqrt(s2) : -1.0;
}

// ---- descending rank of every column, lists of finished and null columns (one workgroup per block)
__global__ void __launch_bounds__(1024) cl_rank_kernel(const Blk* __restrict__ blks)
{
    const Blk d = blks[blockIdx.x];
    for (int c = threadIdx.x; c < d.Np; c += blockDim.x) {
        const double s = d.sig[c];
        int rk = 0;
        for (int o = 0; o < d.Np; ++o) {
            const double so = d.sig[o];
            rk += (so > s) || (so == s && o < c);
        }
        d.rank[c] = rk;
    }
    if (threadIdx.x == 0) {
        const double thresh = sqrt(d.stat[3]);
        int nn = 0, nk = 0;
        for (int c = 0; c < d.N; ++c) {
            if (d.mode == 0 && !(d.sig[c] > thresh)) d.nul[nn++] = c;
            else d.kept[nk++] = c;
        }
        if (nn & 1) d.nul[nn] = -1; // the sub-problem of the null columns needs an even count
        d.cnt[0] = nn;
        d.cnt[1] = nk;
    }
}

// ---- finished columns -> unit vectors; null columns -> pseudo-random candidates / unit vectors after completion
__global__ void __launch_bounds__(NT) cl_cols_kernel(const Blk* __restrict__ blks, int what)
{
    const Blk d = blks[blockIdx.y];
    int c = blockIdx.x;
    if (d.mode == 1) return; // eigh: W is not an output
    if (what == 0) { // normalise the finished columns
        if (c >= d.cnt[1]) return;
        c = d.kept[c];
        const double inv = 1.0 / d.sig[c];
        d2* w = d.W + (int64_t)c * d.M;
        for (int i = threadIdx.x; i < d.M; i += NT) w[i] *= inv;
    } else if (what == 1) { // candidates (splitmix64 of (row, column): any generic vectors do)
        if (c >= d.cnt[0]) return;
        c = d.nul[c];
        d2* w = d.W + (int64_t)c * d.M;
        for (int i = threadIdx.x; i < d.M; i += NT) {
            uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + (uint64_t)(c + 1) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            w[i] = d2{(double)(uint32_t)z * (2.0 / 4294967296.0) - 1.0, (double)(uint32_t)(z >> 32) * (2.0 / 4294967296.0) - 1.0};
        }
    } else { // normalise the completed columns by the norms in cn
        if (c >= d.cnt[0]) return;
        c = d.nul[c];
        const double inv = 1.0 / d.cn[c];
        d2* w = d.W + (int64_t)c * d.M;
        for (int i = threadIdx.x; i < d.M; i += NT) w[i] *= inv;
    }
}

// ---- P[i, j] = <finished column i, candidate j>   (16 x 16 results per workgroup, rows in chunks of 64)
__global__ void __launch_bounds__(NT) cl_dots_kernel(const Blk* __restrict__ blks)
{
    __shared__ d2 As[16][65], Bs[16][65];
    const Blk d = blks[blockIdx.z];
    const int nn = d.cnt[0], nk = d.cnt[1];
    const int ti = blockIdx.y * 16, tj = blockIdx.x * 16;
    if (ti >= nk || tj >= nn) return;
    const int tid = threadIdx.x, ii = tid >> 4, jj = tid & 15;
    d2 acc = d2{0.0, 0.0};
    for (int r0 = 0; r0 < d.M; r0 += 64) {
        for (int e = tid; e < 1024; e += NT) {
            const int col = e >> 6, row = e & 63;
            d2 a = d2{0.0, 0.0}, b = d2{0.0, 0.0};
            if (r0 + row < d.M) {
                if (ti + col < nk) a = d.W[(int64_t)d.kept[ti + col] * d.M + r0 + row];
                if (tj + col < nn) b = d.W[(int64_t)d.nul[tj + col] * d.M + r0 + row];
            }
            As[col][row] = a;
            Bs[col][row] = b;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 64; ++k) acc += cmulc(As[ii][k], Bs[jj][k]);
        __syncthreads();
    }
    if (ti + ii < nk && tj + jj < nn) d.P[(int64_t)(ti + ii) * nn + tj + jj] = acc;
}

// ---- candidate j -= sum_i finished column i * P[i, j]   (64 rows x 16 candidates per workgroup)
__global__ void __launch_bounds__(NT) cl_sub_kernel(const Blk* __restrict__ blks)
{
    __shared__ d2 As[16][64], Ps[16][16];
    const Blk d = blks[blockIdx.z];
    const int nn = d.cnt[0], nk = d.cnt[1];
    const int r0 = blockIdx.x * 64, tj = blockIdx.y * 16;
    if (r0 >= d.M || tj >= nn) return;
    const int tid = threadIdx.x, row = tid & 63, jg = tid >> 6;
    d2 acc[4] = {d2{0.0, 0.0}, d2{0.0, 0.0}, d2{0.0, 0.0}, d2{0.0, 0.0}};
    for (int i0 = 0; i0 < nk; i0 += 16) {
        for (int e = tid; e < 1024; e += NT) {
            const int col = e >> 6, rr = e & 63;
            d2 a = d2{0.0, 0.0};
            if (i0 + col < nk && r0 + rr < d.M) a = d.W[(int64_t)d.kept[i0 + col] * d.M + r0 + rr];
            As[col][rr] = a;
        }
        {
            const int a = tid >> 4, bcol = tid & 15;
            d2 pv = d2{0.0, 0.0};
            if (i0 + a < nk && tj + bcol < nn) pv = d.P[(int64_t)(i0 + a) * nn + tj + bcol];
            Ps[a][bcol] = pv;
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const d2 x = As[a][row];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += cmul(x, Ps[a][jg * 4 + j]);
        }
        __syncthreads();
    }
    if (r0 + row < d.M)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (tj + jg * 4 + j < nn) d.W[(int64_t)d.nul[tj + jg * 4 + j] * d.M + r0 + row] -= acc[j];
}

// ---- results in the reference's order
__global__ void __launch_bounds__(NT) cl_write_kernel(const Blk* __restrict__ blks)
{
    const Blk d = blks[blockIdx.y];
    const double unscl = 1.0 / d.stat[0], shift = d.stat[2];
    const int64_t N = d.N, M = d.M;
    const int64_t stride = (int64_t)gridDim.x * NT;
    const int64_t start = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (d.mode == 1) {
        for (int64_t c = start; c < N; c += stride) d.S[N - 1 - d.rank[c]] = (d.sig[c] - shift) * unscl;
        for (int64_t e = start; e < N * N; e += stride) {
            const int64_t j = e / N, c = e - j * N;
            d.U[j * d.ldu + (N - 1 - d.rank[c])] = d.V[c * d.Np + j];
        }
        return;
    }
    for (int64_t c = start; c < N; c += stride) d.S[d.rank[c]] = fmax(d.sig[c], 0.0) * unscl;
    if (d.tall) { // A = (W / sigma) S V^H
        for (int64_t e = start; e < N * M; e += stride) {
            const int64_t r = e / N, c = e - r * N;
            d.U[r * d.ldu + d.rank[c]] = d.W[c * M + r];
        }
        for (int64_t e = start; e < N * N; e += stride) {
            const int64_t c = e / N, j = e - c * N;
            const d2 v = d.V[c * d.Np + j];
            d.Vh[(int64_t)d.rank[c] * d.ldvh + j] = d2{v.x, -v.y};
        }
    } else { // A^H = (W / sigma) S V^H  =>  A = V S (W / sigma)^H
        for (int64_t e = start; e < N * N; e += stride) {
            const int64_t j = e / N, c = e - j * N;
            d.U[j * d.ldu + d.rank[c]] = d.V[c * d.Np + j];
        }
        for (int64_t e = start; e < N * M; e += stride) {
            const int64_t c = e / M, r = e - c * M;
            const d2 v = d.W[c * M + r];
            d.Vh[(int64_t)d.rank[c] * d.ldvh + r] = d2{v.x, -v.y};
        }
    }
}


size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// Jacobi sweeps over the problems in `jac` until every one has converged; sweeps[i] += sweeps used (-1: no convergence)
int jacobi_sweeps(cyb_ctx_t ctx, std::vector<Jac>& jac, uint32_t* d_off, std::vector<int32_t>& sweeps)
{
    const int np = (int)jac.size();
    std::vector<uint32_t> off((size_t)np);
    std::vector<float> prev((size_t)np, 2.0f);
    for (int sweep = 0; sweep < MAX_SWEEPS; ++sweep) {
        int base = 0, rounds = 0;
        for (Jac& j : jac) {
            j.pair_base = base;
            if (j.active) {
                base += j.ncol / 2;
                rounds = std::max(rounds, j.ncol - 1);
            }
        }
        if (base == 0) return CYB_OK;
        void* d_jac = nullptr;
        CYB_TRY(ctx->upload(jac.data(), sizeof(Jac) * jac.size(), &d_jac));
        CYB_HIP(hipMemsetAsync(d_off, 0, sizeof(uint32_t) * (size_t)np, ctx->stream));
        for (int r = 0; r < rounds; ++r)
            hipLaunchKernelGGL(cl_round_kernel, dim3((unsigned)base), dim3(NT), 0, ctx->stream, static_cast<const Jac*>(d_jac), np, r);
        CYB_HIP(hipGetLastError());
        CYB_HIP(hipMemcpyAsync(off.data(), d_off, sizeof(uint32_t) * (size_t)np, hipMemcpyDeviceToHost, ctx->stream));
        CYB_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < np; ++i) {
            if (!jac[(size_t)i].active) continue;
            ++sweeps[(size_t)i];
            float f;
            std::memcpy(&f, &off[(size_t)i], sizeof(float));
            // the measure was taken BEFORE each pair's rotation: at 1e-10 the rotations just applied leave the columns
            // orthogonal to rounding -- IF the iteration is in its quadratic regime, which a multiple singular value spoils
            // (rotations inside the cluster pass the couplings to the other columns on instead of annihilating them: the
            // real engine left 1.6e-10 behind that way, jacobi_engine.hip kPredictQuad).  So the early exit needs a quadratic
            // step behind it; otherwise the sweeps go on until the measure itself is at the rounding level.
            const float pv = prev[(size_t)i];
            if (f <= 1e-13f || (f <= 1e-10f && pv < 1.0f && f <= 16.0f * pv * pv)) jac[(size_t)i].active = 0;
            prev[(size_t)i] = f;
        }
    }
    int st = CYB_OK;
    for (size_t i = 0; i < jac.size(); ++i)
        if (jac[i].active) {
            sweeps[i] = -1;
            st = CYB_ERR_NOCONV;
        }
    if (st != CYB_OK) cyb::set_error("complex Jacobi (large blocks): a block did not converge in %d sweeps", MAX_SWEEPS);
    return st;
}

} // namespace

int cyb_clarge::run(cyb_ctx_t ctx, const std::vector<Req>& req, int32_t* sweeps_out)
{
    const int nb = (int)req.size();
    if (nb == 0) return CYB_OK;
    // ---- workspace layout
    std::vector<Blk> blk((size_t)nb);
    size_t w0 = 0, w1 = 0, w2 = 0;
    std::vector<size_t> oW((size_t)nb), oV((size_t)nb), oS((size_t)nb), oP((size_t)nb);
    int maxNp = 0;
    int64_t maxM = 0, max_elems = 0;
    for (int i = 0; i < nb; ++i) {
        const Req& r = req[(size_t)i];
        Blk& b = blk[(size_t)i];
        b.tall = r.m >= r.n;
        b.M = (int32_t)std::max(r.m, r.n);
        b.N = (int32_t)std::min(r.m, r.n);
        b.Np = (b.N + 1) & ~1;
        b.m = r.m;
        b.n = r.n;
        b.mode = r.mode;
        b.pad = 0;
        oW[(size_t)i] = w0;
        w0 += align_up(sizeof(d2) * (size_t)b.Np * (size_t)b.M);
        oV[(size_t)i] = w0;
        w0 += align_up(sizeof(d2) * (size_t)b.Np * (size_t)b.Np);
        oS[(size_t)i] = w1;
        // stat (8 doubles), sig, cn (Np doubles each), rank, kept, nul (Np + 2 ints each), cnt (2 ints)
        w1 += align_up(sizeof(double) * (8 + 2 * (size_t)b.Np) + sizeof(int32_t) * (3 * ((size_t)b.Np + 2) + 2));
        oP[(size_t)i] = w2;
        w2 += align_up(sizeof(d2) * ((size_t)b.N / 2 + 1) * ((size_t)b.N / 2 + 1));
        maxNp = std::max(maxNp, (int)b.Np);
        maxM = std::max<int64_t>(maxM, b.M);
        max_elems = std::max<int64_t>(max_elems, (int64_t)b.Np * (b.M + b.Np));
    }
    void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
    CYB_TRY(ctx->workspace(w0, &p0, 0));
    CYB_TRY(ctx->workspace(w1, &p1, 1));
    CYB_TRY(ctx->workspace(w2, &p2, 2));
    CYB_TRY(ctx->workspace(sizeof(uint32_t) * (size_t)nb, &p3, 3));
    for (int i = 0; i < nb; ++i) {
        const Req& r = req[(size_t)i];
        Blk& b = blk[(size_t)i];
        b.A = reinterpret_cast<const d2*>(r.A);
        b.U = reinterpret_cast<d2*>(r.U);
        b.Vh = reinterpret_cast<d2*>(r.Vh);
        b.S = r.S;
        b.lda = r.lda;
        b.ldu = r.ldu;
        b.ldvh = r.ldvh;
        b.W = reinterpret_cast<d2*>(static_cast<char*>(p0) + oW[(size_t)i]);
        b.V = reinterpret_cast<d2*>(static_cast<char*>(p0) + oV[(size_t)i]);
        char* s = static_cast<char*>(p1) + oS[(size_t)i];
        b.stat = reinterpret_cast<double*>(s);
        b.sig = b.stat + 8;
        b.cn = b.sig + b.Np;
        b.rank = reinterpret_cast<int32_t*>(b.cn + b.Np);
        b.kept = b.rank + b.Np + 2;
        b.nul = b.kept + b.Np + 2;
        b.cnt = b.nul + b.Np + 2;
        b.P = reinterpret_cast<d2*>(static_cast<char*>(p2) + oP[(size_t)i]);
    }
    // the descriptor image lives in a slot of the upload ring, which the per-sweep uploads of the Jacobi phases recycle:
    // it is uploaded again after each of them
    void* d_blk = nullptr;
    CYB_TRY(ctx->upload(blk.data(), sizeof(Blk) * blk.size(), &d_blk));
    const Blk* dblk = static_cast<const Blk*>(d_blk);
    uint32_t* d_off = static_cast<uint32_t*>(p3);
    const unsigned chunks = (unsigned)std::min<int64_t>(2048, (max_elems + NT - 1) / NT);
    hipLaunchKernelGGL(cl_prep_kernel, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, dblk);
    hipLaunchKernelGGL(cl_load_kernel, dim3(chunks, (unsigned)nb), dim3(NT), 0, ctx->stream, dblk);
    CYB_HIP(hipGetLastError());
    // ---- phase 1: orthogonalise the columns of W, accumulate V
    std::vector<Jac> jac((size_t)nb);
    std::vector<int32_t> sweeps((size_t)nb, 0);
    for (int i = 0; i < nb; ++i) {
        const Blk& b = blk[(size_t)i];
        jac[(size_t)i] = Jac{b.W, b.V, nullptr, d_off + i, b.stat, b.M, b.Np, b.Np, 0, b.Np >= 2 ? 1 : 0, 0};
    }
    int st = jacobi_sweeps(ctx, jac, d_off, sweeps);
    if (sweeps_out)
        for (int i = 0; i < nb; ++i) sweeps_out[i] = sweeps[(size_t)i];
    if (st != CYB_OK) return st;
    // ---- singular values, order, null columns
    CYB_TRY(ctx->upload(blk.data(), sizeof(Blk) * blk.size(), &d_blk));
    dblk = static_cast<const Blk*>(d_blk);
    hipLaunchKernelGGL(cl_norms_kernel, dim3((unsigned)maxNp, (unsigned)nb), dim3(NT), 0, ctx->stream, dblk, 0);
    hipLaunchKernelGGL(cl_rank_kernel, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, dblk);
    CYB_HIP(hipGetLastError());
    bool any_svd = false;
    for (const Blk& b : blk) any_svd |= (b.mode == 0);
    if (any_svd) {
        std::vector<int32_t> cnt((size_t)nb * 2, 0);
        for (int i = 0; i < nb; ++i)
            CYB_HIP(hipMemcpyAsync(&cnt[(size_t)i * 2], blk[(size_t)i].cnt, sizeof(int32_t) * 2, hipMemcpyDeviceToHost, ctx->stream));
        CYB_HIP(hipStreamSynchronize(ctx->stream));
        int max_null = 0, max_kept = 0;
        for (int i = 0; i < nb; ++i)
            if (blk[(size_t)i].mode == 0) {
                max_null = std::max(max_null, cnt[(size_t)i * 2]);
                max_kept = std::max(max_kept, cnt[(size_t)i * 2 + 1]);
            }
        const Blk* dsvd = dblk; // (eigh blocks of a mixed list have no null columns; cl_cols_kernel skips them)
        if (max_kept > 0)
            hipLaunchKernelGGL(cl_cols_kernel, dim3((unsigned)max_kept, (unsigned)nb), dim3(NT), 0, ctx->stream, dsvd, 0);
        if (max_null > 0) {
            hipLaunchKernelGGL(cl_cols_kernel, dim3((unsigned)max_null, (unsigned)nb), dim3(NT), 0, ctx->stream, dsvd, 1);
            if (max_kept > 0) {
                const dim3 gd((unsigned)((max_null + 15) / 16), (unsigned)((max_kept + 15) / 16), (unsigned)nb);
                const dim3 gs((unsigned)((maxM + 63) / 64), (unsigned)((max_null + 15) / 16), (unsigned)nb);
                for (int pass = 0; pass < 2; ++pass) {
                    hipLaunchKernelGGL(cl_dots_kernel, gd, dim3(NT), 0, ctx->stream, dsvd);
                    hipLaunchKernelGGL(cl_sub_kernel, gs, dim3(NT), 0, ctx->stream, dsvd);
                }
            }
            CYB_HIP(hipGetLastError());
            // candidates orthogonal among themselves: Jacobi on the null columns only, no accumulation
            std::vector<Jac> jc;
            std::vector<int32_t> sw2;
            for (int i = 0; i < nb; ++i) {
                const Blk& b = blk[(size_t)i];
                const int nn = (b.mode == 0) ? cnt[(size_t)i * 2] : 0;
                if (nn < 2) continue;
                jc.push_back(Jac{b.W, nullptr, b.nul, d_off + (int)jc.size(), b.stat, b.M, b.Np, (nn + 1) & ~1, 0, 1, 0});
                sw2.push_back(0);
            }
            if (!jc.empty()) {
                st = jacobi_sweeps(ctx, jc, d_off, sw2);
                if (st != CYB_OK) return st;
                CYB_TRY(ctx->upload(blk.data(), sizeof(Blk) * blk.size(), &d_blk));
                dblk = dsvd = static_cast<const Blk*>(d_blk);
            }
            hipLaunchKernelGGL(cl_norms_kernel, dim3((unsigned)max_null, (unsigned)nb), dim3(NT), 0, ctx->stream, dsvd, 1);
            hipLaunchKernelGGL(cl_cols_kernel, dim3((unsigned)max_null, (unsigned)nb), dim3(NT), 0, ctx->stream, dsvd, 2);
        }
        CYB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(cl_write_kernel, dim3(chunks, (unsigned)nb), dim3(NT), 0, ctx->stream, dblk);
    CYB_HIP(hipGetLastError());
    return CYB_OK;
}
