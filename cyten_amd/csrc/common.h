// Internal shared definitions of libcyten_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <initializer_list>
#include <vector>

#include "../../include/cyten_amd.h"

namespace cyb {

void set_error(const char* fmt, ...);

#define CYB_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            cyb::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),       \
                           __FILE__, __LINE__);                                         \
            return CYB_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define CYB_REQUIRE(cond, ...)                                                          \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            cyb::set_error(__VA_ARGS__);                                                \
            return CYB_ERR_INVALID;                                                     \
        }                                                                               \
    } while (0)

#define CYB_TRY(call)                                                                   \
    do {                                                                                \
        int _s = (call);                                                                \
        if (_s != CYB_OK) return _s;                                                    \
    } while (0)

} // namespace cyb

// Per-device context. Descriptor arrays of grouped calls travel through a ring of pinned-host /
// device slot pairs so the steady state does no hipMalloc (cdna_hip_programming.md Guideline 9).
struct cyb_ctx_s {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    int lds_bytes = 160 * 1024;
    int64_t hbm_bytes = 0;
    char arch[64] = {0};

    static constexpr int kSlots = 32;
    static constexpr int kEvStride = 8; // an event is recorded with every kEvStride-th upload only (hipEventRecord costs the host 2-3 us)
    struct Slot {
        void* dev = nullptr;
        void* host = nullptr; // pinned
        size_t cap = 0;
        hipEvent_t ev = nullptr; // recorded on the stream when this slot was (re)filled
        hipEvent_t copied = nullptr; // recorded on the copy stream after the H2D copy
        bool ev_valid = false;
    };
    hipStream_t copy_stream = nullptr; // descriptor uploads overlap with the kernels of the main stream
    // Side stream of the look-ahead QR (blocked_qr.hip): created on first use with a CU mask that leaves a few CUs to
    // the main stream, so that the next panel's one-workgroup-per-matrix kernel finds a free CU while the trailing update
    // of the previous panel fills the rest of the chip.  aux_state: 0 not tried, 1 ready, -1 unavailable.
    hipStream_t aux_stream = nullptr;
    int aux_state = 0;
    std::vector<hipEvent_t> ev_pool; // timing-disabled events, reused across calls
    int aux(hipStream_t* out);
    int events(size_t n);            // make sure ev_pool holds at least n events
    Slot slots[kSlots];
    uint64_t n_uploads = 0;
    uint64_t ev_waited = 0; // index (+1) of the newest upload whose event the host has waited for
    // large images (more than kBigBytes) go through a ring of their own (see upload())
    static constexpr size_t kBigBytes = 128 * 1024;
    static constexpr int kBig = 8;
    Slot big[kBig];
    uint64_t n_big = 0;
    size_t slot_cap_max = 0; // largest big slot so far: a big slot that must grow grows to this at once (growing costs a stream sync)

    // Copy `bytes` from host `src` into a ring slot and enqueue the H2D copy on the stream.
    // The device pointer stays valid until kSlots/2 further uploads have been made; a grouped
    // call must therefore launch its consumer kernels before making kSlots/2 - 1 more uploads.
    int upload(const void* src, size_t bytes, void** dev_out);

    // grow-only scratch workspaces (device) for decompositions; independent slots so that a routine
    // can call a helper that needs scratch of its own
    static constexpr int kWork = 5; // 0/1: decomposition pipelines, 2: per-round Jacobi scratch and amax partials, 3: convergence flags, 4: partial W1's of the row-split strips
    void* work[kWork] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t work_cap[kWork] = {0, 0, 0, 0, 0};
    int workspace(size_t bytes, void** out, int slot = 0);
    // pinned landing buffer of small device-to-host reads (scalars of reductions, convergence words): a copy into pageable
    // memory goes through the runtime's own staging and costs 2-3x the latency of one into pinned memory
    static constexpr size_t kReadback = 64 * 1024;
    void* readback = nullptr;
    int d2h(void* dst, const void* src, size_t bytes); // copy + wait on the stream (small reads land in `readback` first)
    // measurement hook (cyb_ctx_time_next_gemm): the next grouped-GEMM launch records these two events on the stream directly
    // around its kernel(s) -- after the descriptor upload -- so that a caller times the KERNEL, as rocprofv3 reports it
    hipEvent_t time_start = nullptr, time_stop = nullptr;
};

namespace cyb {
// Several descriptor arrays of ONE launch in ONE ring slot / one H2D copy (each part 256-B aligned): an upload is an
// in-stream copy of a few microseconds plus ~5 us of host API time, and a DMRG-sized bond update makes hundreds.
struct UploadPart {
    const void* src;
    size_t bytes;
    void** out;
};
inline int upload_packed(cyb_ctx_t ctx, std::initializer_list<UploadPart> parts)
{
    size_t tot = 0;
    for (const auto& p : parts) tot = (tot + 255) / 256 * 256 + p.bytes;
    if (tot == 0) {
        for (const auto& p : parts) *p.out = nullptr;
        return CYB_OK;
    }
    std::vector<char> img(tot);
    size_t off = 0;
    for (const auto& p : parts) {
        off = (off + 255) / 256 * 256;
        if (p.bytes) memcpy(img.data() + off, p.src, p.bytes);
        off += p.bytes;
    }
    void* d = nullptr;
    int rc = ctx->upload(img.data(), tot, &d);
    if (rc != CYB_OK) return rc;
    off = 0;
    for (const auto& p : parts) {
        off = (off + 255) / 256 * 256;
        *p.out = p.bytes ? static_cast<char*>(d) + off : nullptr;
        off += p.bytes;
    }
    return CYB_OK;
}
} // namespace cyb

namespace cyb {
// optional per-problem left factor: C = alpha * L (A B) + beta * C with L (M x M, M <= 32, element strides rs/cs)
struct GemmPost {
    const double* L;
    int64_t rs, cs;
};
int gemm_launch_async(cyb_ctx_t ctx, const cyb_gemm_prob* probs, int64_t n_probs, const cyb_gemm_seg* segs, int64_t n_segs,
                      const GemmPost* post = nullptr);

// Staged launches: `gemm_stage` validates a block list and appends its device image (descriptors + tile
// queue) to `image` (256-B aligned); after ONE upload of the whole image, `gemm_launch_staged` enqueues the
// launch that reads it at `dev_image + st.offset`.  A blocked QR stages all its panel steps up front: one
// descriptor upload per factorization instead of three per panel step (each upload is an in-stream copy of a
// few microseconds that the dependent kernels wait for).
struct GemmStaged {
    size_t offset = 0;                 // of this launch's image inside the staging buffer
    size_t off_p = 0, off_s = 0, off_t = 0, off_c = 0;
    int64_t n_tiles = 0;
};
int gemm_stage(cyb_ctx_t ctx, const cyb_gemm_prob* probs, int64_t n_probs, const cyb_gemm_seg* segs, int64_t n_segs,
               const GemmPost* post, std::vector<char>& image, GemmStaged& st);
int gemm_launch_staged(cyb_ctx_t ctx, const GemmStaged& st, void* dev_image, hipStream_t stream = nullptr, int n_cu = 0);

// host-side builder of one grouped-GEMM launch
struct GemmBatch {
    std::vector<cyb_gemm_prob> probs;
    std::vector<cyb_gemm_seg> segs;
    std::vector<GemmPost> post; // empty, or one entry per problem
    // C(M x N, row stride ldc) = alpha * A(M x K) B(K x N) + beta * C, operands as strided views
    void add(double* C, int64_t M, int64_t N, int64_t ldc, const double* A, int64_t a_rs, int64_t a_cs, const double* B,
             int64_t b_rs, int64_t b_cs, int64_t K, double alpha, double beta)
    {
        if (M <= 0 || N <= 0) return;
        cyb_gemm_seg s{A, B, K, a_rs, a_cs, b_rs, b_cs};
        cyb_gemm_prob p{C, M, N, ldc, (int32_t)segs.size(), (int32_t)segs.size() + 1, alpha, beta};
        segs.push_back(s);
        probs.push_back(p);
    }
    // same, with the left factor L (strides l_rs / l_cs) applied to the product
    void add_post(double* C, int64_t M, int64_t N, int64_t ldc, const double* A, int64_t a_rs, int64_t a_cs, const double* B,
                  int64_t b_rs, int64_t b_cs, int64_t K, double alpha, double beta, const double* L, int64_t l_rs, int64_t l_cs)
    {
        if (M <= 0 || N <= 0) return;
        post.resize(probs.size(), GemmPost{nullptr, 0, 0});
        add(C, M, N, ldc, A, a_rs, a_cs, B, b_rs, b_cs, K, alpha, beta);
        post.push_back(GemmPost{L, l_rs, l_cs});
    }
    int launch(cyb_ctx_t ctx)
    {
        if (!post.empty()) post.resize(probs.size(), GemmPost{nullptr, 0, 0});
        return gemm_launch_async(ctx, probs.data(), (int64_t)probs.size(), segs.data(), (int64_t)segs.size(),
                                 post.empty() ? nullptr : post.data());
    }
    int stage(cyb_ctx_t ctx, std::vector<char>& image, GemmStaged& st)
    {
        if (!post.empty()) post.resize(probs.size(), GemmPost{nullptr, 0, 0});
        return gemm_stage(ctx, probs.data(), (int64_t)probs.size(), segs.data(), (int64_t)segs.size(),
                          post.empty() ? nullptr : post.data(), image, st);
    }
    bool empty() const { return probs.empty(); }
};
} // namespace cyb

namespace cyb {
// ---- range safety of the decompositions (scaling.hip)
struct MatRef {
    const double* A;
    int64_t lda, m, n;
};
struct ScaleJob { // dst(rows x cols, ldd) = s * src(rows x cols, lds); src == dst scales in place
    const double* src;
    int64_t lds;
    double* dst;
    int64_t ldd, rows, cols;
    double s;
};
int matrix_amax(cyb_ctx_t ctx, const std::vector<MatRef>& mats, std::vector<double>& amax); // synchronises
double range_scale(double amax); // exact power of two that brings amax into [0.5, 1) if it is outside [1e-90, 1e90], else 1
int scale_copy_batched(cyb_ctx_t ctx, const std::vector<ScaleJob>& jobs);
// grid.x of the (grid.x, n lists) helper kernels whose workgroups stride over the rows / tiles of ONE matrix: about 2048
// workgroups per launch, so that a list dominated by one large matrix still fills the chip (64 per matrix left the 824 x 721
// block of the chi=4096 list on 64 workgroups: 60-180 us per helper launch)
inline unsigned helper_grid_x(size_t n_lists)
{
    const size_t g = 2048 / (n_lists ? n_lists : 1);
    return (unsigned)(g < 64 ? 64 : (g > 512 ? 512 : g));
}

} // namespace cyb

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
