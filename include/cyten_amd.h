/*
 * cyten_amd.h -- C-ABI of the MI355X-native block backend for cyten.
 *
 * This is the drop-in boundary: everything cyten's `BlockBackend`
 * (reference: include/cyten/block_backend/block_backend.h:18-500) needs from a
 * device is reachable through the plain-C entry points below.  No torch, no
 * pybind, no C++ types cross this boundary: device pointers are `void*` /
 * `double*` obtained from any HIP allocator (hipMalloc, a torch tensor's
 * data_ptr, ...), sizes are int64_t, and every call returns an int status
 * (0 = ok) with a thread-local message retrievable via cyb_last_error().
 *
 * Design notes:
 *  - The reference API is one-block-at-a-time (matrix_dot / matrix_svd / ...,
 *    block_backend.h:436-472).  The hardware wants whole block lists, so every
 *    hot entry point here is *grouped*: one call = all blocks of one tensor op,
 *    one (or a few) kernel launches.  The single-block reference calls are the
 *    n=1 special case.
 *  - All work is enqueued on the context's HIP stream and is asynchronous with
 *    respect to the host unless stated otherwise.
 *  - fp64 is the arithmetic type of the hot path (BASELINE.json).  Data
 *    movement entry points are byte-size generic (elem_size 1..16).
 *  - DTYPE RESTRICTION (deliberate).  The reference's Dtype set is bool, int64,
 *    float32, complex64, float64, complex128 (include/cyten/block_backend/
 *    dtypes.h:12-21).  Arithmetic entry points exist for float64 (`_f64`) and
 *    complex128 (`_c128`: interleaved (re, im) storage, numpy's layout); bool
 *    is a 1-byte storage type for masks and comparison results
 *    (cyb_compare_f64, cyb_count_nonzero_u8, cyb_convert_u8_f64).  float32,
 *    complex64 and int64 blocks have NO arithmetic here: they can be moved
 *    (cyb_copy_strided_batched with elem_size 4 / 8) but the host mirror
 *    (HipBlockBackend.to_dtype) refuses them, and the Array-API namespace of
 *    integration/ keeps such data -- index arrays from argsort / argmin, int64
 *    scalars -- on the host.  Nothing on the tdot / SVD / QR / eigh path of
 *    cyten uses them.
 *  - A change of the context's stream (cyb_ctx_set_stream) orders the new
 *    stream behind everything enqueued on the old one (the workspaces and the
 *    descriptor ring are shared).
 */
#ifndef CYTEN_AMD_H
#define CYTEN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CYB_VERSION 100 /* 0.1.0 */

/* status codes */
enum {
    CYB_OK = 0,
    CYB_ERR_INVALID = 1,  /* bad argument (-> std::invalid_argument / ValueError in the adapter) */
    CYB_ERR_HIP = 2,      /* a HIP runtime call failed */
    CYB_ERR_NOCONV = 3,   /* an iterative decomposition did not converge (-> LinAlgError) */
    CYB_ERR_NOMEM = 4,
    CYB_ERR_UNSUPPORTED = 5
};

typedef struct cyb_ctx_s* cyb_ctx_t;

/* ---- context / stream / errors ------------------------------------------------------------ */

/* Version of the library (CYB_VERSION it was built with). */
int cyb_version(void);
/* Thread-local message of the last failing call ("" if none). */
const char* cyb_last_error(void);
/* Create a context on HIP device `device`. `stream` is a hipStream_t (may be NULL = the
 * device's null stream); the context does not own it.
 * Mirrors the per-device backend singleton of the reference
 * (src/block_backend/torch.cpp:669-695, numpy.cpp:411-432). */
int cyb_ctx_create(cyb_ctx_t* out, int device, void* stream);
int cyb_ctx_destroy(cyb_ctx_t ctx);
int cyb_ctx_set_stream(cyb_ctx_t ctx, void* stream);
/* Block the host until everything enqueued on the context's stream is done.
 * Replaces BlockBackend::synchronize (block_backend.h:478-479). */
int cyb_ctx_sync(cyb_ctx_t ctx);
/* Device properties the host planner needs. */
int cyb_device_info(cyb_ctx_t ctx, int* n_cu, int* lds_bytes, int64_t* hbm_bytes, char* arch, int arch_len);

/* ---- raw memory (used by hosts that do not bring their own allocator) ---------------------- */
int cyb_malloc(cyb_ctx_t ctx, void** out, size_t bytes);
int cyb_free(cyb_ctx_t ctx, void* ptr);
int cyb_memcpy_h2d(cyb_ctx_t ctx, void* dst, const void* src, size_t bytes); /* async on stream */
int cyb_memcpy_d2h(cyb_ctx_t ctx, void* dst, const void* src, size_t bytes); /* synchronous */
int cyb_memcpy_d2d(cyb_ctx_t ctx, void* dst, const void* src, size_t bytes); /* async */
int cyb_memset(cyb_ctx_t ctx, void* dst, int byte, size_t bytes);            /* async */

/* ---- timing (HIP events on the context's stream) ------------------------------------------- */
typedef struct cyb_event_s* cyb_event_t;
int cyb_event_create(cyb_event_t* out);
int cyb_event_destroy(cyb_event_t ev);
int cyb_event_record(cyb_ctx_t ctx, cyb_event_t ev);
int cyb_event_elapsed_ms(cyb_event_t start, cyb_event_t stop, float* ms); /* syncs on stop */
/* Measurement hook: the NEXT asynchronous grouped-GEMM launch of this context (cyb_gemm_grouped_enqueue_f64,
 * cyb_compose_plan_enqueue_f64) records `start` / `stop` on the stream directly around its kernel(s), i.e. behind the
 * descriptor upload -- the kernel duration rocprofv3 reports, not the enqueue.  One shot; NULL events clear it. */
int cyb_ctx_time_next_gemm(cyb_ctx_t ctx, cyb_event_t start, cyb_event_t stop);

/* ---- grouped block GEMM (tdot hot loop) ------------------------------------------------------
 * Replaces the loop  block = bb.matrix_dot(a,b); block = block + bb.matrix_dot(a',b'); ...
 * of abelian_compose_worker (reference src/backends/abelian.cpp:1424-1460), one
 * NumpyBlockBackend::matrix_dot = np.dot per pair (src/block_backend/numpy.cpp:1218-1225),
 * and FusionTreeBackend::compose (src/backends/fusion_tree_backend.cpp:669-698).
 *
 * One *problem* is one result block  C (M x N, row stride ldc, unit column stride)
 *      C = alpha * sum_{s in segments} A_s(M x K_s) * B_s(K_s x N)  + beta * C
 * The segments are the K-split pairs the reference accumulates with Block::operator+; here
 * they are accumulated in registers inside one tile pass (no extra C traffic).
 * Operands are strided *views*: element (i,k) of A_s is A[i*a_rs + k*a_cs]; one of the two
 * strides must be 1 (row- or column-major view, i.e. permuted/transposed blocks need no copy).
 */
typedef struct {
    const double* A;
    const double* B;
    int64_t K;
    int64_t a_rs, a_cs; /* element strides of A: rows (M index), cols (K index) */
    int64_t b_rs, b_cs; /* element strides of B: rows (K index), cols (N index) */
} cyb_gemm_seg;

typedef struct {
    double* C;
    int64_t M, N;
    int64_t ldc;       /* row stride of C in elements (>= N) */
    int32_t seg_begin; /* segments [seg_begin, seg_end) of the seg array belong to this problem */
    int32_t seg_end;
    double alpha, beta; /* beta == 0: C is not read */
} cyb_gemm_prob;

typedef struct cyb_gemm_plan_s* cyb_gemm_plan_t;

/* Build a launch plan (tile queue + device-resident descriptors) for a block list. Host arrays
 * are copied; pointers inside them must stay valid device pointers while the plan is run. */
int cyb_gemm_plan_create(cyb_ctx_t ctx, cyb_gemm_plan_t* out,
                         const cyb_gemm_prob* probs, int64_t n_probs,
                         const cyb_gemm_seg* segs, int64_t n_segs);
int cyb_gemm_plan_run(cyb_ctx_t ctx, cyb_gemm_plan_t plan);
int cyb_gemm_plan_destroy(cyb_gemm_plan_t plan);
/* algorithmic flops (sum 2*M*N*K) and bytes (8*(MK+KN+MN)) of the plan, number of launches/tiles */
int cyb_gemm_plan_info(cyb_gemm_plan_t plan, double* flops, double* bytes, int64_t* n_tiles, int32_t* n_launches);
/* convenience: create + run + destroy (synchronises the device) */
int cyb_gemm_grouped_f64(cyb_ctx_t ctx,
                         const cyb_gemm_prob* probs, int64_t n_probs,
                         const cyb_gemm_seg* segs, int64_t n_segs);
/* One-shot asynchronous form: validates, stages the descriptors through the context's pinned upload
 * ring and enqueues the launch on the context's stream -- no plan object, no device allocation, no
 * host synchronisation.  This is what a block list that is contracted once (every compose of a Krylov
 * matvec, abelian.cpp:1424-1460) should use; the operands must stay alive until the stream reaches
 * the launch (stream-ordered allocators give that for free). */
int cyb_gemm_grouped_enqueue_f64(cyb_ctx_t ctx,
                                 const cyb_gemm_prob* probs, int64_t n_probs,
                                 const cyb_gemm_seg* segs, int64_t n_segs);
/* Back-to-back v_mfma_f64_16x16x4_f64 issue micro-benchmark: returns measured TFLOP/s of the
 * chip (every CU issuing, `iters` MFMAs per wave on independent accumulators held in AGPRs;
 * waves_per_simd = accumulators * 100 + waves, accumulators 1/2/4/8, default 4).  The measured fp64
 * MFMA ceiling next to the spec one: 74.8 of 78.6 TFLOP/s on MI355X. */
int cyb_mfma_f64_peak(cyb_ctx_t ctx, int iters, int waves_per_simd, double* tflops, double* ms);

/* ---- batched per-block decompositions ----------------------------------------------------------
 * One descriptor per sector block; the whole block list of a tensor goes in one call.
 * All matrices are row-major (C order, as numpy blocks are) with explicit row strides.
 */

/* Thin SVD  A(m x n) = U(m x k) diag(S(k)) Vh(k x n),  k = min(m,n),  S descending, >= 0.
 * Replaces NumpyBlockBackend::matrix_svd = scipy.linalg.svd(a, full_matrices=False)
 * (src/block_backend/numpy.cpp:1247-1297), called per block from AbelianBackend::svd
 * (src/backends/abelian.cpp:3517-3518) and FusionTreeBackend::svd (fusion_tree_backend.cpp:2211).
 * A is not modified. */
typedef struct {
    const double* A; int64_t lda;
    int64_t m, n;
    double* U; int64_t ldu;   /* m x k */
    double* S;                /* k */
    double* Vh; int64_t ldvh; /* k x n */
} cyb_svd_desc;
/* info[i] (host array, may be NULL): number of Jacobi sweeps used for block i, or <0 if it did
 * not converge within max_sweeps (then the call returns CYB_ERR_NOCONV).  The call synchronises
 * the stream when info != NULL. */
int cyb_svd_batched_f64(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* info);

/* The same with options, for the caller that TRUNCATES afterwards (truncated_svd_py, /root/reference/src/tensors/
 * decompositions.cpp:673-712: svd -> truncate_singular_values -> svd_apply_mask discards every vector that is not kept).
 *   flags & CYB_SVD_SKIP_NULL_VECTORS: singular vectors of numerically zero singular values (sigma below
 *       |A|_F max(m,n) eps, numpy.linalg.matrix_rank's threshold) are not completed to an orthonormal set: the
 *       corresponding columns of U / rows of Vh are unspecified (zero on the normalised side).  S and all other vectors
 *       are exactly those of cyb_svd_batched_f64.  Saves the second blocked QR of every rank-deficient block -- every
 *       block of a two-site theta = A.B is rank-deficient by construction.
 *   rank[i] (host array, may be NULL): number of singular values of block i above that threshold; a caller that keeps
 *       more than rank[i] values of a block must call cyb_svd_batched_f64 for it instead. */
#define CYB_SVD_SKIP_NULL_VECTORS 1
/* every block is the interleaved real embedding M(A) of a complex block: entry a + ib -> [[a, -b], [b, a]], 2m x 2n, both
 * extents even.  The factors come back embedded the same way (U: 2m x 2k, S: 2k with every value twice, Vh: 2k x 2n): real
 * column 2a of U IS complex column a (rows 0::2 real parts, rows 1::2 imaginary parts), real row 2a of Vh IS complex row a
 * (columns 0::2 real parts, columns 1::2 minus the imaginary parts), and the two belong to the same singular triplet.  Where
 * the block is rank deficient or graded over many decades the real factors are orthogonal but structured only up to
 * eps * sigma_max / sigma_j: a caller that needs complex orthonormality there re-orthonormalises those columns
 * (cyten_amd/block_backend.py, _complex_svd_embedded).  CYB_ERR_UNSUPPORTED: the list has too many rows for one persistent
 * sweep launch (use cyb_svd_batched_c128).  numpy.cpp:1247-1297 for complex128 blocks, on the float64 block engine
 * (DESIGN.md section 4.5b). */
#define CYB_SVD_EMBEDDED_COMPLEX 2
int cyb_svd_batched_ex_f64(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* info, int32_t flags, int32_t* rank);

/* QR  A(m x n) = Q R.  economic: Q m x k, R k x n (k=min(m,n)); full: Q m x m, R m x n.
 * Replaces NumpyBlockBackend::matrix_qr = scipy.linalg.qr(a, mode=...) (numpy.cpp:1236-1245);
 * matrix_lq (block_backend.cpp:1033-1040) is built on it by the host.
 * Sign convention as LAPACK dgeqrf (R diagonal may be negative). */
typedef struct {
    const double* A; int64_t lda;
    int64_t m, n;
    double* Q; int64_t ldq;
    double* R; int64_t ldr;
    int32_t full;
} cyb_qr_desc;
int cyb_qr_batched_f64(cyb_ctx_t ctx, const cyb_qr_desc* descs, int64_t n);

/* Hermitian (real symmetric) eigendecomposition  A(n x n) = V diag(W) V^T,  W ascending.
 * Replaces NumpyBlockBackend::eigh = np.linalg.eigh (numpy.cpp:658-680). Only the lower
 * triangle convention of LAPACK is not relied upon: A is assumed symmetric.
 * V may be NULL (eigvalsh, numpy.cpp:682-698). */
typedef struct {
    const double* A; int64_t lda;
    int64_t n;
    double* W;
    double* V; int64_t ldv;
} cyb_eigh_desc;
int cyb_eigh_batched_f64(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* info);
/* flags & CYB_EIGH_EMBEDDED_COMPLEX: every block is the interleaved real embedding M(H) (2n x 2n, n even in the embedding's
 * extents) of a complex Hermitian block: W comes back with every eigenvalue twice (2n values), real column 2a of V is the
 * complex eigenvector a (rows 0::2 real parts, rows 1::2 imaginary parts) -- exactly structured, no QR step is involved.
 * CYB_ERR_UNSUPPORTED as for the SVD.  np.linalg.eigh of complex128 blocks (numpy.cpp:658-680) on the float64 block engine. */
#define CYB_EIGH_EMBEDDED_COMPLEX 2
int cyb_eigh_batched_ex_f64(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* info, int32_t flags);

/* ---- data movement: strided N-d copies (permute_axes / reshape-copy / get_item / set_item,
 *      combine_legs / split_legs sub-block scatter/gather) ------------------------------------
 * dst[sum_d i_d*dst_strides[d]] = src[sum_d i_d*src_strides[d]]  for i_d in [0, shape[d]).
 * Strides in elements of elem_size bytes (1,2,4,8,16). ndim <= CYB_MAX_NDIM. Replaces the numpy
 * views of numpy.cpp:924-931 (permute_axes), :1057-1064 (reshape) once a contiguous result is
 * needed, and the `new_block[slices] = combined` scatter of abelian.cpp:1212-1214 /
 * `old_block[slices]` gather of abelian.cpp:3414-3427.  The whole list is one launch. */
#define CYB_MAX_NDIM 8
typedef struct {
    void* dst;
    const void* src;
    int32_t ndim;
    int32_t conj; /* 1: complex-conjugate while copying (elem_size 16 = complex128 only) */
    int64_t shape[CYB_MAX_NDIM];
    int64_t dst_strides[CYB_MAX_NDIM];
    int64_t src_strides[CYB_MAX_NDIM];
} cyb_copy_desc;
int cyb_copy_strided_batched(cyb_ctx_t ctx, const cyb_copy_desc* descs, int64_t n, int32_t elem_size);

/* ---- BLAS-1 class block-list ops (Lanczos / truncation callers: norm, inner,
 *      linear_combination, mul, scale_axis; numpy.cpp:898-913, :815-842, :1358-1385) ----------- */
typedef struct {
    const double* x; /* contiguous */
    const double* y; /* contiguous, may be NULL where unused */
    double* out;     /* contiguous, may alias x or y, may be NULL where unused */
    int64_t n;
} cyb_vec_desc;
/* result[0] = sum over all list entries of sum_i x_i*y_i  (y == NULL: x_i*x_i); device scalar */
int cyb_dot_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double* result_dev);
/* per-entry results: result_dev[j] = sum_i x_i*y_i of entry j */
int cyb_dot_each_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double* result_dev);
/* out = a*x + b*y  for every entry (y may be NULL with b ignored) */
int cyb_axpby_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double a, double b);
/* result_dev[0] = max_i |x_i| over the whole list */
int cyb_maxabs_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, double* result_dev);
/* elementwise binary op on contiguous lists: out = x (op) y; op: 0 add, 1 sub, 2 mul, 3 div, 4 pow (Block::pow(Block), numpy.cpp power) */
int cyb_binary_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, int32_t op);
/* elementwise unary op: out = f(x) (out may be x); op: 0 abs, 1 sqrt, 2 exp, 3 log, 4 neg, 5 square, 6 reciprocal,
 * 7 round to float32 (the cast-on-store of float32 / complex64 blocks, dtypes.h:12-21: they are held in double words),
 * 8 truncate toward zero (int64 blocks) */
int cyb_unary_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, int32_t op);

/* elementwise unary op with one parameter: op 0 cutoff_inverse (|x| < param ? 0 : 1/x, numpy.cpp:645-656),
 * 1 stable_log (x > param ? log x : 0, numpy.cpp:1088-1098), 2 pow (x ** param, Block::pow numpy.cpp:265-270),
 * 3 angle of a real number (numpy.cpp:587-594; param ignored) */
int cyb_unary_param_batched_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, int32_t op, double param);
/* out[i] = x[i] (op) y[i]  (y == NULL: x[i] (op) scalar) as one byte per element, 0 / 1: the boolean blocks of
 * Block::operator< <= > >= == != (numpy.cpp:229-263); op: 0 lt, 1 le, 2 gt, 3 ge, 4 eq, 5 ne.  Contiguous. */
int cyb_compare_f64(cyb_ctx_t ctx, const double* x, const double* y, double scalar, uint8_t* out, int64_t n, int32_t op);
/* out[i] = x[i] ? 1.0 : 0.0  (to_dtype of a boolean block, numpy.cpp:1131-1138) */
int cyb_convert_u8_f64(cyb_ctx_t ctx, const uint8_t* x, double* out, int64_t n);
/* result_dev[0] = number of non-zero bytes: any / all / sum_all of a boolean block (numpy.cpp:568-575, 596-603) */
int cyb_count_nonzero_u8(cyb_ctx_t ctx, const uint8_t* x, int64_t n, uint64_t* result_dev);
/* extremum of a contiguous vector together with its flat index, ties resolved to the lowest index like np.argmax /
 * np.argmin: mode 0 max (numpy.cpp:871-878), 1 min (:889-896, argmin :552-566), 2 max |x| (abs_argmax :533-550).
 * result_dev[0] = the maximised key (x, -x or |x|), result_dev[1] = the index as an int64 bit pattern. */
int cyb_extremum_f64(cyb_ctx_t ctx, const double* x, int64_t n, int32_t mode, double* result_dev);

/* out[i, j, k] = x[i, j, k] * f[j]  for a block viewed as (outer, axis, inner), contiguous.
 * Replaces scale_axis (numpy.cpp:1373-1385). */
typedef struct {
    const double* x;
    const double* f;
    double* out;
    int64_t outer, axis, inner;
} cyb_scale_axis_desc;
int cyb_scale_axis_batched_f64(cyb_ctx_t ctx, const cyb_scale_axis_desc* descs, int64_t n);

/* Gather (apply_mask, numpy.cpp:605-613) / scatter-into-zeros (enlarge_leg, numpy.cpp:700-728)
 * along one axis of a block viewed as (outer, axis, inner).  idx (device, int64) lists the kept
 * positions (n_keep entries).  gather: out(outer,n_keep,inner) = x(outer, idx[j], inner);
 * scatter: out(outer,axis,inner) = 0 except out(:, idx[j], :) = x(:, j, :). */
typedef struct {
    const double* x;
    double* out;
    const int64_t* idx;
    int64_t outer, axis, inner, n_keep;
} cyb_mask_desc;
int cyb_mask_gather_batched_f64(cyb_ctx_t ctx, const cyb_mask_desc* descs, int64_t n);
int cyb_mask_scatter_batched_f64(cyb_ctx_t ctx, const cyb_mask_desc* descs, int64_t n);

/* ---- complex128 on the tdot path ------------------------------------------------------------------
 * The reference's blocks are float64 or complex128 (Dtype, include/cyten/block_backend/dtypes.h; numpy.cpp
 * dispatches every virtual on it).  Complex blocks are stored interleaved (re, im) as numpy does.  A complex
 * product runs through the same real grouped GEMM: A is read in place as a real M x 2K matrix, C written in place
 * as a real M x 2N matrix, and B is expanded once into the real 2K x 2N matrix [[br, bi], [-bi, br]] per element
 * (cyb_complex_expand_batched_f64).  Decompositions of complex blocks: the *_c128 entries further down. */
typedef struct cyb_cexpand_desc {
    const double* src; /* complex K x N view, element (k, n) at src + 2*(k*rs + n*cs) */
    int64_t rs, cs;    /* strides in complex elements */
    int64_t K, N;
    double* dst;       /* 2K x 2N doubles, row-major, contiguous */
} cyb_cexpand_desc;
int cyb_complex_expand_batched_f64(cyb_ctx_t ctx, const cyb_cexpand_desc* descs, int64_t n);
/* out = a*x + b*y on complex vectors of desc.n complex elements (y may be NULL): Block::operator+ / mul /
 * linear_combination for complex128 (numpy.cpp:1358-1365) */
/* elementwise functions of complex vectors (desc.n complex elements; x, y interleaved complex): op 0 |z| and
 * 4 angle write desc.n doubles to out; 1 sqrt, 2 exp, 3 log, 5 z*w, 6 z/w write complex
 * (abs / sqrt / exp / log / angle numpy.cpp:449-456,1066-1073,730-737,862-869,587-594; Block::operator* /  :217-227) */
int cyb_elementwise_batched_c128(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n, int32_t op);
int cyb_axpby_batched_c128(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n,
                           double a_re, double a_im, double b_re, double b_im);

/* ---- host-side sector matching of a contraction (no device work) ------------------------------------------------
 * The int64 bookkeeping of abelian_compose_worker (src/backends/abelian.cpp:1239-1469) in C++, as in the reference:
 * contracts the last `num_contr` legs of a with the first `num_contr` legs of b (a.legs[na_legs - 1 - i] pairs with
 * b.legs[i]).  Block tables are row-major int64 (one row per block, one column per leg, entries = sector index on that
 * leg); a leg is its sorted sector charges (n_sectors x n_sym), multiplicities and orientation sign; moduli[k] = 0 for a
 * U(1) factor, N for Z_N.  The plan lists, per result block (rows lexsorted with the last column as primary key, like
 * BlockInds::lexsort_indices), its sector indices, its shape and the (a-block, b-block) pairs whose products are summed:
 * pairs of result g are pair_a/pair_b[group_offsets[g] .. group_offsets[g + 1]).  flops = sum 2 M N K. */
typedef struct {
    int64_t n_sectors;
    const int64_t* sectors; /* n_sectors x n_sym */
    const int64_t* mults;   /* n_sectors */
    int32_t sign;           /* +1 / -1 */
    int32_t pad;
} cyb_leg;
typedef struct cyb_compose_plan_s* cyb_compose_plan_t;
int cyb_compose_plan_create(const int64_t* moduli, int32_t n_sym, const cyb_leg* a_legs, int32_t na_legs,
                            const int64_t* a_block_inds, int64_t na_blocks, const cyb_leg* b_legs, int32_t nb_legs,
                            const int64_t* b_block_inds, int64_t nb_blocks, int32_t num_contr, cyb_compose_plan_t* out);
int cyb_compose_plan_sizes(cyb_compose_plan_t plan, int64_t* n_res, int64_t* n_pairs, int64_t* n_cols);
int cyb_compose_plan_get(cyb_compose_plan_t plan, int64_t* res_block_inds, int64_t* res_shapes, int64_t* group_offsets,
                         int64_t* pair_a, int64_t* pair_b, double* flops);
/* The hot loop of abelian_compose_worker (src/backends/abelian.cpp:1424-1460: matrix_dot + operator+ per matched pair,
 * reshape of the result) for operand blocks that are C-contiguous float64 arrays: the descriptors of the grouped GEMM are
 * built inside the library from the plan and three address tables, and ONE asynchronous launch is enqueued on the context's
 * stream.  a_ptrs[i] / b_ptrs[j]: device addresses of a's i-th / b's j-th block (the block-table order the plan was created
 * with; with more than one contracted leg the b-blocks must hold their contracted axes in a's reversed order, as
 * abelian.cpp:1349-1382 permutes them).  which: indices of the plan's result blocks to compute (n_which of them; NULL = all,
 * in plan order) -- a rank of a sharded contraction computes its own sectors only; out_ptrs[k]: address of the M x N result
 * of which[k].  flops / bytes (may be NULL): algorithmic 2 M N K and 8 (M K + K N + M N) sums of what was enqueued. */
int cyb_compose_plan_enqueue_f64(cyb_ctx_t ctx, cyb_compose_plan_t plan, const int64_t* a_ptrs, const int64_t* b_ptrs,
                                 const int64_t* which, int64_t n_which, const int64_t* out_ptrs, double* flops, double* bytes);
int cyb_compose_plan_destroy(cyb_compose_plan_t plan);

/* ---- complex128 decompositions -----------------------------------------------------------------------------------
 * Same descriptors as the float64 entries, every matrix pointer addressing interleaved (re, im) storage and every
 * leading dimension counted in complex elements; S and W stay real.  Blocks with min(m, n) <= 64, max(m, n) <= 128 and
 * 32 * (Np * (max | 1) + Np * (Np | 1)) <= 150 KB (Np = min rounded up to even; e.g. 64 x 64, 48 x 96, 40 x 128): one
 * workgroup per block runs a complex one-sided Jacobi iteration in LDS (csrc/csvd_small.hip).  Larger blocks: the same
 * iteration with the work matrix in device memory, one launch per tournament round for all blocks of the call, null
 * directions of rank-deficient blocks completed block-wise (csrc/csvd_large.hip; plain FMA arithmetic, not the MFMA
 * block engine of the float64 path).  info[i] = sweeps used.
 * NumpyBlockBackend::matrix_svd / eigh on complex128 blocks (numpy.cpp:1247-1297, 658-680). */
int cyb_svd_batched_c128(cyb_ctx_t ctx, const cyb_svd_desc* descs, int64_t n, int32_t* info);
int cyb_eigh_batched_c128(cyb_ctx_t ctx, const cyb_eigh_desc* descs, int64_t n, int32_t* info);
/* QR of complex128 blocks (scipy.linalg.qr economic / full, numpy.cpp:1236-1245: LAPACK zgeqrf + zungqr): Householder
 * reflections built as zlarfg builds them (csrc/cqr_house.hip) -- one workgroup per block for blocks of at most 96 x 96
 * elements, one launch per column step for the whole list beyond.  Backward stable for every block (rank deficient,
 * copied or zero columns, graded); Q unitary, R upper triangular with a real non-negative diagonal (row j of R and column
 * j of Q carry the sign of LAPACK's beta_j, so scipy's R agrees up to the signs of its rows). */
int cyb_qr_batched_c128(cyb_ctx_t ctx, const cyb_qr_desc* descs, int64_t n);

/* ---- linear combinations of strided views ------------------------------------------------
 * dst[idx] = (accumulate ? dst[idx] : 0) + sum_{t in [term_begin, term_end)} coeff_t * src_t[idx]  for idx over `shape`,
 * all strides in elements.  One launch for the tree-block updates of FusionTreeBackend::apply_instructions
 * (src/backends/fusion_tree_backend.cpp:593-631): TreePairMapping::transform_tensor
 * (src/backends/fusion_tree_mapping.cpp:391-513) builds `tree_block = sum_I f_JI * old_block[slice_I]`
 * (:447-468: get_item, mul, Block::operator+), applies permute_combined_matrix (:491-492) and stores
 * `new_block[slices] = permuted` (:493-497) once per (tree pair, term); here the permutation lives in the source
 * strides and the placement in the destination view.  Destinations of one call must not overlap. */
typedef struct {
    double* dst;
    int32_t ndim;
    int32_t accumulate;
    int32_t term_begin, term_end;
    int64_t shape[CYB_MAX_NDIM];
    int64_t dst_strides[CYB_MAX_NDIM];
} cyb_lincomb_desc;
typedef struct {
    const double* src;
    double coeff;
    int64_t src_strides[CYB_MAX_NDIM];
} cyb_lincomb_term;
int cyb_lincomb_strided_batched_f64(cyb_ctx_t ctx, const cyb_lincomb_desc* descs, int64_t n, const cyb_lincomb_term* terms,
                                    int64_t n_terms);
/* complex128 form of the same: `dst` and the sources are interleaved (re, im) arrays, shapes / strides count complex
 * elements, coefficients are complex -- every anyonic R / C / B symbol is (tests/python_tests/backends/
 * test_fusion_tree_backend.py:59-71).  A term with src_real != 0 reads a float64 source (strides in doubles): real data
 * under a complex mapping, the `dtype = to_complex(dtype)` of fusion_tree_mapping.cpp:433-436. */
typedef struct {
    const double* src;
    double coeff_re, coeff_im;
    int32_t src_real;
    int32_t reserved;
    int64_t src_strides[CYB_MAX_NDIM];
} cyb_lincomb_term_c128;
int cyb_lincomb_strided_batched_c128(cyb_ctx_t ctx, const cyb_lincomb_desc* descs, int64_t n, const cyb_lincomb_term_c128* terms,
                                     int64_t n_terms);

/* ---- truncation of singular values on the device -----------------------------------------
 * TensorBackend::_truncate_singular_values_selection (src/backends/tensor_backend.cpp:139-242) applied to the
 * concatenation of the per-sector singular values WITHOUT the host round trip of
 * AbelianBackend::truncate_singular_values (src/backends/abelian.cpp:3623-3638, to_numpy of all S at :3631).
 * descs[s].x / .n: the singular values of sector s (device, contiguous; at most 65536 values in total, else
 * CYB_ERR_UNSUPPORTED; up to 8192 the sort runs entirely in LDS).  Outputs (device): keep_idx_dev[off_s + j] = ascending positions (within sector s) of the kept
 * values, off_s = sum of the n of the sectors before s -- the index tables of cyb_mask_gather_batched_f64;
 * mask_dev[off_s + i] = 1 if value i of sector s is kept; result_dev[0] = err (sum of the discarded S^2),
 * result_dev[1] = new_norm (sum of the kept S^2), result_dev[2 + s] = kept count of sector s as an int64 bit pattern. */
typedef struct {
    int64_t chi_max;        /* keep at most chi_max values; < 0: no limit */
    int64_t chi_min;        /* keep at least chi_min values (>= 1) */
    double degeneracy_tol;  /* do not cut between values with log(S[i]/S[i-1]) < degeneracy_tol; 0: off */
    double trunc_cut;       /* discard while the discarded weight stays <= trunc_cut^2 */
    double svd_min;         /* keep only S >= svd_min (if has_svd_min) */
    int32_t has_svd_min;
    int32_t minimize_error; /* 1: smallest admissible cut (reference default), 0: largest */
} cyb_trunc_opts;
int cyb_truncate_select_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n_sectors, const cyb_trunc_opts* opts,
                            int64_t* keep_idx_dev, uint8_t* mask_dev, double* result_dev);
/* The same with the marginal error of a value weighted by the quantum dimension of its sector, err_i = d_s * S_i^2 -- the
 * `qdims` argument of tensor_backend.cpp:158-164, which FusionTreeBackend::truncate_singular_values fills with ONE number
 * per coupled sector (fusion_tree_backend.cpp:2280-2303).  sector_weights: n_sectors positive numbers on the HOST, or NULL
 * (all 1: the abelian case).  err and new_norm are the weighted sums. */
int cyb_truncate_select_weighted_f64(cyb_ctx_t ctx, const cyb_vec_desc* descs, int64_t n_sectors, const double* sector_weights,
                                     const cyb_trunc_opts* opts, int64_t* keep_idx_dev, uint8_t* mask_dev, double* result_dev);

/* fill: out[i] = value (zeros / ones_block); eye: out (n x n, contiguous) = identity
 * (eye_matrix, numpy.cpp:1197-1207) */
int cyb_fill_f64(cyb_ctx_t ctx, double* out, int64_t n, double value);
int cyb_eye_f64(cyb_ctx_t ctx, double* out, int64_t n);
/* counter-based standard-normal fill (random_normal; Philox4x32-10 + Box-Muller), sigma-scaled */
int cyb_random_normal_f64(cyb_ctx_t ctx, double* out, int64_t n, uint64_t seed, double sigma);
/* counter-based uniform fill on [lo, hi) (random_uniform, numpy.cpp:965-988 draws from [-1, 1)) */
int cyb_random_uniform_f64(cyb_ctx_t ctx, double* out, int64_t n, uint64_t seed, double lo, double hi);

#ifdef __cplusplus
}
#endif
#endif /* CYTEN_AMD_H */
