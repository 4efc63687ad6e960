// Microbenchmark of ONE step of the in-LDS 32x32 two-sided Jacobi eigensolver (development aid; not product code).
// Variants isolate the pieces of a step: LDS round trip, rotation chains, 2x2 updates, scattered stores, barrier.
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/probes/eig_step_probe scripts/probes/eig_step_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int JP = 32, GS = 34, VS = 34, NT = 256;

__device__ __forceinline__ int ring_next(int p)
{
    if (p == 0) return 0;
    if (p == 1) return 2;
    if (p & 1) return p - 2;
    return p == JP - 2 ? JP - 1 : p + 2;
}
__device__ __forceinline__ void rot_bf(double a, double d, double b, double& c, double& s)
{
    const double delta = d - a;
    const double b2 = b + b;
    double h2 = fma(b2, b2, delta * delta);
    const bool ok = (fabs(b) > 1e-300) & (h2 > 1e-300);
    h2 = ok ? h2 : 1.0;
    double r = __builtin_amdgcn_rsq(h2);
    double g = h2 * r, h = 0.5 * r;
    double e = fma(-g, h, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-g, h, 0.5);
    h = fma(h, e, h);
    const double c2 = fma(fabs(delta), h, 0.5);
    double r2 = __builtin_amdgcn_rsq(c2);
    double gc = c2 * r2, hc = 0.5 * r2;
    double ec = fma(-gc, hc, 0.5);
    gc = fma(gc, ec, gc);
    hc = fma(hc, ec, hc);
    ec = fma(-gc, hc, 0.5);
    gc = fma(gc, ec, gc);
    hc = fma(hc, ec, hc);
    const double sabs = (fabs(b) * h) * (4.0 * hc);
    const bool pos = (delta >= 0.0) == (b >= 0.0);
    c = ok ? gc : 1.0;
    s = ok ? (pos ? sabs : -sabs) : 0.0;
}
// f32 seed variant: rsq in f32, three Goldschmidt steps in f64
__device__ __forceinline__ double rsq_seed32(double x) { return (double)__builtin_amdgcn_rsqf((float)x); }

__device__ __forceinline__ double dpp_d(double old, double src, const int ctrl_is_shl)
{
    const int slo = __double2loint(src), shi = __double2hiint(src), olo = __double2loint(old), ohi = __double2hiint(old);
    int rlo, rhi;
    if (ctrl_is_shl) {
        rlo = __builtin_amdgcn_update_dpp(olo, slo, 0x101, 0xf, 0xf, false); // row_shl:1  lane i <- lane i + 1
        rhi = __builtin_amdgcn_update_dpp(ohi, shi, 0x101, 0xf, 0xf, false);
    } else {
        rlo = __builtin_amdgcn_update_dpp(olo, slo, 0x111, 0xf, 0xf, false); // row_shr:1  lane i <- lane i - 1
        rhi = __builtin_amdgcn_update_dpp(ohi, shi, 0x111, 0xf, 0xf, false);
    }
    return __hiloint2double(rhi, rlo);
}
// ring step on a (top, bot) pair of column positions held by the 16 lanes of a row: top moves right, bot moves left
__device__ __forceinline__ void ring_dpp(double& top, double& bot, int pc)
{
    const double t_shr = dpp_d(top, top, 0);   // lane 0 keeps its own top (position 0 is fixed)
    const double b_shr = dpp_d(0.0, bot, 0);   // lane 1 receives bot[0]
    const double b_shl = dpp_d(top, bot, 1);   // lane 15 receives its own top (the turn of the ring)
    top = (pc == 1) ? b_shr : t_shr;
    bot = b_shl;
}

// VAR: 0 full step; 1 no rotation maths (constants); 2 rotations only (no LDS traffic except the reads); 3 one chain only;
//      4 full step without V; 5 full, no barrier (WRONG results, timing only); 6 V in registers, moved by DPP
template <int VAR>
__global__ void __launch_bounds__(NT, 1) probe(const double* gin, double* gout, unsigned long long* cyc, int reps)
{
    __shared__ __attribute__((aligned(16))) double Gsm[2 * JP * GS + 2 * JP * VS];
    double* Ga = Gsm;
    double* Gb = Ga + JP * GS;
    double* Vc = Gb + JP * GS;
    double* Vn = Vc + JP * VS;
    const int tid = threadIdx.x;
    for (int e = tid; e < JP * JP; e += NT) {
        Ga[(e / JP) * GS + e % JP] = gin[e];
        Vc[(e / JP) * VS + e % JP] = (e / JP == e % JP) ? 1.0 : 0.0;
    }
    __syncthreads();
    const int pr = tid >> 4, pc = tid & 15;
    const int r0 = 2 * pr, c0 = 2 * pc;
    const int dr0 = ring_next(r0), dr1 = ring_next(r0 + 1), dc0 = ring_next(c0), dc1 = ring_next(c0 + 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = wall_clock64();
    double sink = 0.0;
    double w00 = (r0 == c0) ? 1.0 : 0.0, w01 = 0.0, w10 = 0.0, w11 = (r0 == c0) ? 1.0 : 0.0; // VAR 6: V block in registers
    for (int rep = 0; rep < reps; ++rep) {
        for (int r = 0; r < JP - 1; ++r) {
            const d2 g0 = *reinterpret_cast<const d2*>(Ga + r0 * GS + c0);
            const d2 g1 = *reinterpret_cast<const d2*>(Ga + (r0 + 1) * GS + c0);
            const d2 ar = *reinterpret_cast<const d2*>(Ga + r0 * GS + r0);
            const double dr = Ga[(r0 + 1) * GS + r0 + 1];
            const d2 ac = *reinterpret_cast<const d2*>(Ga + c0 * GS + c0);
            const double dc = Ga[(c0 + 1) * GS + c0 + 1];
            d2 v0 = d2{0, 0}, v1 = d2{0, 0};
            if (VAR != 4 && VAR != 6) {
                v0 = *reinterpret_cast<const d2*>(Vc + r0 * VS + c0);
                v1 = *reinterpret_cast<const d2*>(Vc + (r0 + 1) * VS + c0);
            }
            double c1 = 0.8, s1 = 0.6, c2 = 0.6, s2 = 0.8;
            if (VAR == 7) { // ONE chain per thread (its column pair); the row pair's rotation comes from lane pc == pr of the 16-lane row
                rot_bf(ac.x, dc, ac.y, c2, s2);
                const int src = ((tid & 63) & 48) | pr;
                c1 = __shfl(c2, src);
                s1 = __shfl(s2, src);
            } else if (VAR != 1) {
                rot_bf(ar.x, dr, ar.y, c1, s1);
                if (VAR != 3) rot_bf(ac.x, dc, ac.y, c2, s2);
            }
            const double hik = c1 * g0.x - s1 * g1.x, hil = c1 * g0.y - s1 * g1.y;
            const double hjk = s1 * g0.x + c1 * g1.x, hjl = s1 * g0.y + c1 * g1.y;
            double nik = c2 * hik - s2 * hil, nil = s2 * hik + c2 * hil;
            double njk = c2 * hjk - s2 * hjl, njl = s2 * hjk + c2 * hjl;
            if (pr == pc) {
                nil = 0.0;
                njk = 0.0;
            }
            if (VAR == 2) {
                sink += nik + nil + njk + njl;
            } else {
                Gb[dr0 * GS + dc0] = nik;
                Gb[dr0 * GS + dc1] = nil;
                Gb[dr1 * GS + dc0] = njk;
                Gb[dr1 * GS + dc1] = njl;
                if (VAR == 6) {
                    double t0 = c2 * w00 - s2 * w01, b0 = s2 * w00 + c2 * w01;
                    double t1 = c2 * w10 - s2 * w11, b1 = s2 * w10 + c2 * w11;
                    ring_dpp(t0, b0, pc);
                    ring_dpp(t1, b1, pc);
                    w00 = t0;
                    w01 = b0;
                    w10 = t1;
                    w11 = b1;
                } else if (VAR != 4) {
                    Vn[r0 * VS + dc0] = c2 * v0.x - s2 * v0.y;
                    Vn[r0 * VS + dc1] = s2 * v0.x + c2 * v0.y;
                    Vn[(r0 + 1) * VS + dc0] = c2 * v1.x - s2 * v1.y;
                    Vn[(r0 + 1) * VS + dc1] = s2 * v1.x + c2 * v1.y;
                }
            }
            if (VAR != 5) __syncthreads();
            double* t = Ga;
            Ga = Gb;
            Gb = t;
            t = Vc;
            Vc = Vn;
            Vn = t;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    __syncthreads();
    if (tid == 0) {
        cyc[2 * blockIdx.x] = t1 - t0;
        cyc[2 * blockIdx.x + 1] = w1 - w0;
    }
    for (int e = tid; e < JP * JP; e += NT) gout[blockIdx.x * JP * JP + e] = Ga[(e / JP) * GS + e % JP] + sink * 1e-300;
    if (VAR == 6) { // V^T G0 V must be diagonal: hand V back through gout's tail for the host check
        double* vo = gout + (size_t)gridDim.x * JP * JP + (size_t)blockIdx.x * JP * JP;
        vo[r0 * JP + c0] = w00;
        vo[r0 * JP + c0 + 1] = w01;
        vo[(r0 + 1) * JP + c0] = w10;
        vo[(r0 + 1) * JP + c0 + 1] = w11;
    }
}

int main()
{
    const int nb = 24, reps = 20;
    std::vector<double> h(JP * JP);
    srand(1);
    std::vector<double> x(JP * 64);
    for (auto& v : x) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < JP; ++i)
        for (int j = 0; j < JP; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += x[i * 64 + k] * x[j * 64 + k];
            h[i * JP + j] = s;
        }
    double *gin, *gout;
    unsigned long long* cyc;
    hipMalloc(&gin, sizeof(double) * JP * JP);
    hipMalloc(&gout, sizeof(double) * JP * JP * nb * 2);
    hipMalloc(&cyc, sizeof(unsigned long long) * 2 * nb);
    hipMemcpy(gin, h.data(), sizeof(double) * JP * JP, hipMemcpyHostToDevice);
    const char* names[8] = {"full step", "no rotation maths", "rotations only, no stores", "one rotation chain", "full without V", "full without barrier", "V in registers (DPP)", "one chain + lane exchange"};
    for (int var = 0; var < 8; ++var) {
        for (int it = 0; it < 3; ++it) {
            switch (var) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(NT), 0, 0, gin, gout, cyc, reps); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(nb), dim3(NT), 0, 0, gin, gout, cyc, reps); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(nb), dim3(NT), 0, 0, gin, gout, cyc, reps); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(nb), dim3(NT), 0, 0, gin, gout, cyc, reps); break;
                case 4: hipLaunchKernelGGL(probe<4>, dim3(nb), dim3(NT), 0, 0, gin, gout, cyc, reps); break;
                case 5: hipLaunchKernelGGL(probe<5>, dim3(nb), dim3(NT), 0, 0, gin, gout, cyc, reps); break;
                case 6: hipLaunchKernelGGL(probe<6>, dim3(nb), dim3(NT), 0, 0, gin, gout, cyc, reps); break;
                case 7: hipLaunchKernelGGL(probe<7>, dim3(nb), dim3(NT), 0, 0, gin, gout, cyc, reps); break;
            }
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> hc(2 * nb);
        hipMemcpy(hc.data(), cyc, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost);
        double mean = 0, wall = 0;
        for (int b = 0; b < nb; ++b) {
            mean += (double)hc[2 * b];
            wall += (double)hc[2 * b + 1];
        }
        mean /= nb;
        wall /= nb;
        // s_memtime: shader cycles; wall_clock64: 100 MHz
        printf("[eigprobe] %-28s %8.1f s_memtime ticks per step, %6.1f ns per step\n", names[var], mean / (reps * 31.0), wall / (reps * 31.0) * 10.0);
    }
    std::vector<double> ho(JP * JP);
    hipMemcpy(ho.data(), gout, sizeof(double) * JP * JP, hipMemcpyDeviceToHost);
    double off = 0, dia = 0;
    for (int i = 0; i < JP; ++i)
        for (int j = 0; j < JP; ++j) (i == j ? dia : off) += ho[i * JP + j] * ho[i * JP + j];
    printf("[eigprobe] after the last variant: off/diag = %.3e\n", off / dia);
    {   // variant 6: V^T G0 V diagonal and V orthogonal?
        std::vector<double> hv(JP * JP);
        hipMemcpy(hv.data(), gout + (size_t)nb * JP * JP, sizeof(double) * JP * JP, hipMemcpyDeviceToHost);
        double offv = 0, diav = 0, orth = 0;
        for (int i = 0; i < JP; ++i)
            for (int j = 0; j < JP; ++j) {
                double s = 0, o = 0;
                for (int k = 0; k < JP; ++k) {
                    double t = 0;
                    for (int l = 0; l < JP; ++l) t += h[k * JP + l] * hv[l * JP + j];
                    s += hv[k * JP + i] * t;
                    o += hv[k * JP + i] * hv[k * JP + j];
                }
                (i == j ? diav : offv) += s * s;
                orth = fmax(orth, fabs(o - (i == j ? 1.0 : 0.0)));
            }
        printf("[eigprobe] DPP variant: |offdiag(V^T G V)|^2 / |diag|^2 = %.3e, max |V^T V - 1| = %.3e\n", offv / diav, orth);
    }
    return 0;
}
