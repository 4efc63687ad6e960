// One-sided block-Jacobi engine shared by the batched SVD and eigh entry points.
//
// Working set per matrix ("vectors are rows"):
//   W : nvp x lenp row-major, the nv vectors to orthogonalise (rows), zero padded
//   J : nvp x nvp  row-major accumulated transform (starts as identity), optional
// After convergence  W = T * W0  has mutually orthogonal rows (norms = singular values) and
// J = T.  See svd_jacobi.hip for how U/S/Vh (or eigenpairs) are read off.
#pragma once
#include "common.h"

namespace cyb {

constexpr int JB = 16;      // vectors per block
constexpr int JP = 2 * JB;  // vectors per pair problem (Gram is JP x JP)

struct JMat {
    double* W;
    double* J;      // may be nullptr: no accumulation (eigh of a shifted, well conditioned matrix)
    int32_t nvp;    // padded vector count, multiple of 64 (so that J rows are whole 64-column chunks)
    int32_t lenp;   // padded vector length, multiple of 64
    int32_t nb;     // nvp / JB (even)
    int32_t nv;     // true vector count
    int32_t len;    // true vector length
    int32_t pad;
    double tol;     // convergence threshold on |g_ij| / sqrt(g_ii g_jj)
    double thr2;    // rows with 0 < |w|^2 <= thr2 are numerically null and get zeroed (deflation); 0: off
};

struct JWork {
    int32_t mat;
    int32_t slot;
};

// Host driver: runs sweeps over all matrices until converged (or max_sweeps).  `h_mats` are the
// host copies of the descriptors (device copy made inside).  sweeps_out[i] = sweeps used, or
// -1 if matrix i did not converge.  Synchronises the stream once per sweep (reads a few bytes).
int jacobi_orthogonalise(cyb_ctx_t ctx, const std::vector<JMat>& h_mats, int max_sweeps,
                         std::vector<int32_t>& sweeps_out, bool cplx = false);

} // namespace cyb
