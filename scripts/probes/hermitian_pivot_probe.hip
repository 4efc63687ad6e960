// Unit probe of hermitian_pivot_solve (jacobi_engine.hip): one wave diagonalises M(G_c) of a random 16 x 16 Hermitian
// Gram matrix; the host checks that the returned 32 x 32 matrix is orthogonal, structured and diagonalises the input.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Icyten_amd/csrc scripts/probes/hermitian_pivot_probe.hip \
//         -Lcyten_amd/lib -lcyten_amd -Wl,-rpath,$PWD/cyten_amd/lib -o gpurun_out/hpp
#include "../../cyten_amd/csrc/jacobi_engine.hip"
#include <cstdio>
#include <random>
#include <vector>

namespace cyb {
namespace {
__global__ void __launch_bounds__(64, 1) probe_kernel(const double* G, double* V, double* lam, int sweeps)
{
    __shared__ double Gs[JP * GS], work[4 * CJ * CS], rot[64], Vo[JP * VS];
    const int lane = threadIdx.x;
    for (int e = lane; e < JP * JP; e += 64) Gs[(e / JP) * GS + e % JP] = G[e];
    __syncthreads();
    hermitian_pivot_solve(Gs, work, rot, Vo, Gs, sweeps, lane);
    __syncthreads();
    for (int e = lane; e < JP * JP; e += 64) V[e] = Vo[(e / JP) * VS + e % JP];
    if (lane < JP) lam[lane] = Gs[lane * GS + lane];
}
} // namespace
} // namespace cyb

int main()
{
    using namespace cyb;
    const int n = 40;
    std::mt19937_64 rng(3);
    std::normal_distribution<double> nd;
    std::vector<double> xr(CJ * n), xi(CJ * n), G(JP * JP, 0.0);
    for (auto& v : xr) v = nd(rng);
    for (auto& v : xi) v = nd(rng);
    // M(X): row 2a = [x, -y], row 2a+1 = [y, x]
    std::vector<double> M(JP * 2 * n);
    for (int a = 0; a < CJ; ++a)
        for (int j = 0; j < n; ++j) {
            M[(2 * a) * 2 * n + 2 * j] = xr[a * n + j];
            M[(2 * a) * 2 * n + 2 * j + 1] = -xi[a * n + j];
            M[(2 * a + 1) * 2 * n + 2 * j] = xi[a * n + j];
            M[(2 * a + 1) * 2 * n + 2 * j + 1] = xr[a * n + j];
        }
    for (int i = 0; i < JP; ++i)
        for (int j = 0; j < JP; ++j) {
            double s = 0;
            for (int k = 0; k < 2 * n; ++k) s += M[i * 2 * n + k] * M[j * 2 * n + k];
            G[i * JP + j] = s;
        }
    double *dG, *dV, *dl;
    hipMalloc(&dG, JP * JP * 8);
    hipMalloc(&dV, JP * JP * 8);
    hipMalloc(&dl, JP * 8);
    hipMemcpy(dG, G.data(), JP * JP * 8, hipMemcpyHostToDevice);
    for (int sweeps : {1, 3, 8}) {
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dG, dV, dl, sweeps);
        std::vector<double> V(JP * JP), lam(JP);
        hipMemcpy(V.data(), dV, JP * JP * 8, hipMemcpyDeviceToHost);
        hipMemcpy(lam.data(), dl, JP * 8, hipMemcpyDeviceToHost);
        double orth = 0, offd = 0, dg = 0, st = 0, gmax = 0;
        for (int i = 0; i < JP; ++i)
            for (int j = 0; j < JP; ++j) {
                double s = 0;
                for (int k = 0; k < JP; ++k) s += V[k * JP + i] * V[k * JP + j];
                orth = std::max(orth, std::fabs(s - (i == j)));
                gmax = std::max(gmax, std::fabs(G[i * JP + j]));
            }
        std::vector<double> T(JP * JP);
        for (int i = 0; i < JP; ++i)
            for (int j = 0; j < JP; ++j) {
                double s = 0;
                for (int k = 0; k < JP; ++k) s += G[i * JP + k] * V[k * JP + j];
                T[i * JP + j] = s;
            }
        for (int i = 0; i < JP; ++i)
            for (int j = 0; j < JP; ++j) {
                double s = 0;
                for (int k = 0; k < JP; ++k) s += V[k * JP + i] * T[k * JP + j];
                if (i == j) dg = std::max(dg, std::fabs(s - lam[i]));
                else offd = std::max(offd, std::fabs(s));
            }
        for (int a = 0; a < CJ; ++a)
            for (int b = 0; b < CJ; ++b) {
                st = std::max(st, std::fabs(V[(2 * a) * JP + 2 * b] - V[(2 * a + 1) * JP + 2 * b + 1]));
                st = std::max(st, std::fabs(V[(2 * a) * JP + 2 * b + 1] + V[(2 * a + 1) * JP + 2 * b]));
            }
        printf("sweeps %d: |V^T V - 1| = %.2e  off(V^T G V)/|G| = %.2e  |diag - lam|/|G| = %.2e  structure = %.2e\n", sweeps, orth,
               offd / gmax, dg / gmax, st);
    }
    return 0;
}
