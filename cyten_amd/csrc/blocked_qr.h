// Batched blocked Householder QR on column-major working matrices (internal).
//
//   A = Q R,  Q = H_0 H_1 ... H_{k-1},  H_j = I - tau_j v_j v_j^T  (LAPACK dgeqrf conventions),
//   panels of NBK columns: unblocked Householder inside the panel (one workgroup per matrix),
//   compact-WY block reflector  I - V T V^T  applied to the trailing matrix / to Q's columns with
//   three grouped GEMM launches per panel step for the WHOLE batch.
#pragma once
#include "common.h"

namespace cyb {

constexpr int NBK = 32;

struct BqrMat {
    double* Ac;      // col-major m x n (element (i,c) at Ac[c*ld + i]); overwritten: R in the upper triangle
    int64_t ld;      // >= m
    int32_t m, n, k; // k = min(m,n)
    double* V;       // col-major m x k (same ld): explicit unit-lower-trapezoidal reflectors
    double* T;       // ceil(k/NBK) blocks of NBK x NBK (row-major, upper triangular)
    double* tau;     // k
    double* scratch; // (1 + kWSplit) * scr_half doubles: W2, then the row-chunk partials of W1
    int64_t scr_half; // NBK * max(n, kc_max)
    int32_t v_zeroed = 0; // the caller has zero-filled V (a memset of its workspace): the panel kernels skip the rows above a panel
    // Early stop (SVD preconditioner only): ctl -> 2 zeroed doubles, parts -> ceil(n / NBK) * bqr_strip_slots(m) doubles, stop_rel2 > 0: the
    // factorisation stops once ||A[j:, j:]||_F^2 <= stop_rel2 x (largest trailing norm seen); the remaining reflectors are the
    // identity (T = 0: the caller's workspace must be zeroed), the remaining rows of R are that trailing block's upper
    // triangle, at the rounding level of the matrix.
    double* ctl = nullptr;
    double* parts = nullptr;
    double stop_rel2 = 0.0;
    int32_t reflect_always = 0; // a column whose tail below the pivot is exactly zero is still reflected (H = I - 2 e e^T,
                                // R_jj = -alpha) instead of LAPACK's tau = 0: the diagonal of R then ALWAYS has the sign
                                // -sign(alpha), which keeps the R of an interleaved complex embedding structured (the
                                // last column of a square matrix, whose partner column was reflected)
};

// bytes of V + T + tau + scratch for an m x n matrix whose Q will be applied to at most kc columns
size_t bqr_aux_bytes(int64_t m, int64_t n, int64_t ld, int64_t kc);
// carve V/T/tau/scratch out of `base` (256-B aligned pieces); returns bytes used
size_t bqr_carve(BqrMat& q, char* base, int64_t kc);

// number of row chunks (workgroups) a strip of `rows` rows is split into by the two-launch form of the block-reflector
// application; a factorisation with early stop needs ceil(n / NBK) * bqr_strip_slots(m) doubles of `parts`
int bqr_strip_slots(int64_t rows);

// factor every matrix (asynchronous on ctx->stream)
int bqr_factor(cyb_ctx_t ctx, const std::vector<BqrMat>& mats);

struct BqrTarget {
    int32_t mat;  // index into mats
    double* C;    // col-major m x kc, ld = ldc
    int64_t ldc;
    int32_t kc;
};
// C <- Q C for every target (asynchronous). At most one target per matrix per call (the targets
// share the matrix' scratch), kc <= the kc the matrix was carved for.
int bqr_apply_q(cyb_ctx_t ctx, const std::vector<BqrMat>& mats, const std::vector<BqrTarget>& targets);

// Batched tile transpose  out[r*ldo + c] = in[c*ldi + r]  for r < R, c < C  (row-major <-> col-major).
// upper != 0: entries with r > c or r >= rlim are written as 0 (extraction of R).
// plain != 0: no transpose, out[r*ldo + c] = in[r*ldi + c] (2-D copy with different leading dimensions).
struct XposeDesc {
    const double* in;
    double* out;
    int64_t ldi, ldo;
    int32_t R, C, upper, rlim;
    int32_t plain, pad;
};
int xpose_batched(cyb_ctx_t ctx, const std::vector<XposeDesc>& descs);
// C (col-major m x kc, ld) = first kc columns of the identity
struct EyeDesc {
    double* C;
    int64_t ld;
    int32_t m, kc;
    int32_t col0, pad; // column c of C is the unit vector e_{c + col0}
};
int eye_cols_batched(cyb_ctx_t ctx, const std::vector<EyeDesc>& descs);

} // namespace cyb
