// Context, error, memory and event entry points of the C-ABI (include/cyten_amd.h).
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace cyb {
static thread_local char g_err[1024] = {0};
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
} // namespace cyb

int cyb_ctx_s::upload(const void* src, size_t bytes, void** dev_out)
{
    if (bytes == 0) {
        *dev_out = nullptr;
        return CYB_OK;
    }
    if (bytes > kBigBytes) {
        // Large descriptor images (the panel-step images of the blocked QR, the tile lists of big grouped GEMMs) have a small
        // ring of their own: in the common ring they wandered from slot to slot, and every slot they touched had to grow -- each
        // time behind a stream synchronisation plus hipMalloc / hipHostMalloc (0.45 ms in the middle of a pipeline).  A big
        // slot is reused after kBig further BIG uploads; the event recorded kBig/2 big uploads later was enqueued after its
        // consumers (same contract as below, counted in big uploads).
        const uint64_t jb = n_big;
        Slot& b = big[jb % kBig];
        if (jb >= (uint64_t)kBig) {
            Slot& w = big[(jb - kBig / 2) % kBig];
            if (w.ev_valid) CYB_HIP(hipEventSynchronize(w.ev));
        }
        if (b.cap < bytes) {
            size_t ncap = std::max<size_t>(b.cap ? b.cap : 2 * kBigBytes, slot_cap_max);
            while (ncap < bytes) ncap *= 2;
            slot_cap_max = ncap; // (the next big slot that must grow grows to this at once)
            if (b.dev) {
                CYB_HIP(hipStreamSynchronize(stream)); // old buffer may still be read by an in-flight kernel
                CYB_HIP(hipFree(b.dev));
                CYB_HIP(hipHostFree(b.host));
                b.dev = b.host = nullptr;
                b.cap = 0;
            }
            CYB_HIP(hipMalloc(&b.dev, ncap));
            CYB_HIP(hipHostMalloc(&b.host, ncap, hipHostMallocDefault));
            b.cap = ncap;
        }
        if (!b.ev) CYB_HIP(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
        CYB_HIP(hipEventRecord(b.ev, stream));
        b.ev_valid = true;
        memcpy(b.host, src, bytes);
        CYB_HIP(hipMemcpyAsync(b.dev, b.host, bytes, hipMemcpyHostToDevice, stream));
        *dev_out = b.dev;
        n_big++;
        return CYB_OK;
    }
    const uint64_t j = n_uploads;
    Slot& s = slots[j % kSlots];
    // Before reusing this slot (filled kSlots uploads ago) make sure its consumers are done: they were enqueued before
    // upload j - kSlots/2 (the contract below), so any event recorded at or after that upload was enqueued after them.
    // Events are recorded with every kEvStride-th upload only: the first such index >= j - kSlots/2 is still older than j.
    static_assert(kSlots / 2 > kEvStride, "the event an upload waits for must already have been recorded");
    if (j >= (uint64_t)kSlots) {
        const uint64_t e = (j - kSlots / 2 + kEvStride - 1) / kEvStride * kEvStride;
        if (e + 1 > ev_waited) {
            Slot& w = slots[e % kSlots];
            if (w.ev_valid) CYB_HIP(hipEventSynchronize(w.ev));
            ev_waited = e + 1;
        }
    }
    if (s.cap < bytes) {
        size_t ncap = s.cap ? s.cap : (size_t)1 << 16;
        while (ncap < bytes) ncap *= 2;
        if (s.dev) {
            // old buffer may still be read by an in-flight kernel
            CYB_HIP(hipStreamSynchronize(stream));
            CYB_HIP(hipFree(s.dev));
            CYB_HIP(hipHostFree(s.host));
            s.dev = s.host = nullptr;
            s.cap = 0;
        }
        CYB_HIP(hipMalloc(&s.dev, ncap));
        CYB_HIP(hipHostMalloc(&s.host, ncap, hipHostMallocDefault));
        s.cap = ncap;
    }
    if (j % kEvStride == 0) {
        if (!s.ev) CYB_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
        CYB_HIP(hipEventRecord(s.ev, stream));
        s.ev_valid = true;
    }
    memcpy(s.host, src, bytes);
    // Experiment kept as a knob (off): descriptor lists up to CYB_UPLOAD_ZEROCOPY bytes are read by the kernels straight
    // from the pinned slot (zero copy) instead of being copied.  Measured: no gain where it was meant to help (toy DMRG,
    // chi=256: 0.48-0.49 vs 0.49-0.51 s per sweep) and a loss on the chi=4096 step (52.3 -> 55.6 ms at 2 KB, 54.1 at
    // 64 KB): the round kernels of the decompositions wait for the first descriptor read over the host link.
    static const size_t zero_copy_max = getenv("CYB_UPLOAD_ZEROCOPY") ? (size_t)atoll(getenv("CYB_UPLOAD_ZEROCOPY")) : 0;
    if (bytes <= zero_copy_max) {
        void* alias = nullptr;
        if (hipHostGetDevicePointer(&alias, s.host, 0) == hipSuccess && alias) {
            *dev_out = alias;
            n_uploads++;
            return CYB_OK;
        }
        (void)hipGetLastError();
    }
    // (measured: issuing the copy on a second stream + hipStreamWaitEvent is SLOWER than the in-stream
    // copy -- 76.0 vs 69.6 ms per batched SVD of the chi=4096 list -- so uploads stay in-stream)
    CYB_HIP(hipMemcpyAsync(s.dev, s.host, bytes, hipMemcpyHostToDevice, stream));
    *dev_out = s.dev;
    n_uploads++;
    return CYB_OK;
}

int cyb_ctx_s::d2h(void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return CYB_OK;
    if (bytes <= kReadback) {
        if (!readback) CYB_HIP(hipHostMalloc(&readback, kReadback, hipHostMallocDefault));
        CYB_HIP(hipMemcpyAsync(readback, src, bytes, hipMemcpyDeviceToHost, stream));
        CYB_HIP(hipStreamSynchronize(stream));
        memcpy(dst, readback, bytes);
        return CYB_OK;
    }
    CYB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
    CYB_HIP(hipStreamSynchronize(stream));
    return CYB_OK;
}

int cyb_ctx_s::aux(hipStream_t* out)
{
    if (aux_state == 0) {
        // every 16th CU stays with the main stream (16 of 256: two per XCD)
        const int words = (n_cu + 31) / 32;
        std::vector<uint32_t> mask((size_t)words, 0xffffffffu);
        for (int cu = 0; cu < n_cu; cu += 16) mask[(size_t)cu / 32] &= ~(1u << (cu % 32));
        if (n_cu % 32) mask[(size_t)words - 1] &= (1u << (n_cu % 32)) - 1u;
        aux_state = (hipExtStreamCreateWithCUMask(&aux_stream, (uint32_t)words, mask.data()) == hipSuccess) ? 1 : -1;
        if (aux_state < 0) {
            (void)hipGetLastError();
            aux_stream = nullptr;
        }
    }
    *out = aux_stream;
    return aux_state > 0 ? CYB_OK : CYB_ERR_UNSUPPORTED;
}

int cyb_ctx_s::events(size_t n)
{
    while (ev_pool.size() < n) {
        hipEvent_t e = nullptr;
        CYB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ev_pool.push_back(e);
    }
    return CYB_OK;
}

int cyb_ctx_s::workspace(size_t bytes, void** out, int slot)
{
    if (slot < 0 || slot >= kWork) {
        cyb::set_error("workspace: bad slot %d", slot);
        return CYB_ERR_INVALID;
    }
    if (bytes > work_cap[slot]) {
        if (work[slot]) {
            CYB_HIP(hipStreamSynchronize(stream));
            CYB_HIP(hipFree(work[slot]));
            work[slot] = nullptr;
            work_cap[slot] = 0;
        }
        size_t ncap = bytes + bytes / 4 + (1 << 20);
        CYB_HIP(hipMalloc(&work[slot], ncap));
        work_cap[slot] = ncap;
    }
    *out = work[slot];
    return CYB_OK;
}

extern "C" {

int cyb_version(void) { return CYB_VERSION; }

const char* cyb_last_error(void) { return cyb::g_err; }

int cyb_ctx_create(cyb_ctx_t* out, int device, void* stream)
{
    CYB_REQUIRE(out != nullptr, "cyb_ctx_create: out is NULL");
    int ndev = 0;
    CYB_HIP(hipGetDeviceCount(&ndev));
    CYB_REQUIRE(device >= 0 && device < ndev, "cyb_ctx_create: device %d out of range (%d devices)",
                device, ndev);
    CYB_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    CYB_HIP(hipGetDeviceProperties(&prop, device));
    cyb_ctx_s* c = new cyb_ctx_s();
    c->device = device;
    c->stream = (hipStream_t)stream;
    c->n_cu = prop.multiProcessorCount;
    c->lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    c->hbm_bytes = (int64_t)prop.totalGlobalMem;
    snprintf(c->arch, sizeof(c->arch), "%s", prop.gcnArchName);
    *out = c;
    return CYB_OK;
}

int cyb_ctx_destroy(cyb_ctx_t ctx)
{
    if (!ctx) return CYB_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->slots) {
        if (s.dev) (void)hipFree(s.dev);
        if (s.host) (void)hipHostFree(s.host);
        if (s.ev) (void)hipEventDestroy(s.ev);
        if (s.copied) (void)hipEventDestroy(s.copied);
    }
    for (auto& s : ctx->big) {
        if (s.dev) (void)hipFree(s.dev);
        if (s.host) (void)hipHostFree(s.host);
        if (s.ev) (void)hipEventDestroy(s.ev);
    }
    if (ctx->readback) (void)hipHostFree(ctx->readback);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (auto& w : ctx->work)
        if (w) (void)hipFree(w);
    delete ctx;
    return CYB_OK;
}

int cyb_ctx_set_stream(cyb_ctx_t ctx, void* stream)
{
    CYB_REQUIRE(ctx, "cyb_ctx_set_stream: ctx is NULL");
    hipStream_t next = (hipStream_t)stream;
    if (next != ctx->stream) {
        // The grow-only workspaces, the upload ring's device slots and pooled outputs may still be in use by kernels
        // queued on the old stream: everything launched on the new stream from now on waits for the old stream's tail.
        hipEvent_t ev = nullptr;
        CYB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(next, ev, 0);
        (void)hipEventDestroy(ev); // (destruction is deferred by the runtime until the event has completed)
        CYB_HIP(e);
        ctx->stream = next;
    }
    return CYB_OK;
}

int cyb_ctx_sync(cyb_ctx_t ctx)
{
    CYB_REQUIRE(ctx, "cyb_ctx_sync: ctx is NULL");
    CYB_HIP(hipStreamSynchronize(ctx->stream));
    return CYB_OK;
}

int cyb_device_info(cyb_ctx_t ctx, int* n_cu, int* lds_bytes, int64_t* hbm_bytes, char* arch,
                    int arch_len)
{
    CYB_REQUIRE(ctx, "cyb_device_info: ctx is NULL");
    if (n_cu) *n_cu = ctx->n_cu;
    if (lds_bytes) *lds_bytes = ctx->lds_bytes;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    if (arch && arch_len > 0) snprintf(arch, (size_t)arch_len, "%s", ctx->arch);
    return CYB_OK;
}

int cyb_malloc(cyb_ctx_t ctx, void** out, size_t bytes)
{
    CYB_REQUIRE(ctx && out, "cyb_malloc: NULL argument");
    CYB_HIP(hipSetDevice(ctx->device));
    CYB_HIP(hipMalloc(out, bytes ? bytes : 8));
    return CYB_OK;
}

int cyb_free(cyb_ctx_t ctx, void* ptr)
{
    CYB_REQUIRE(ctx, "cyb_free: ctx is NULL");
    if (ptr) CYB_HIP(hipFree(ptr));
    return CYB_OK;
}

int cyb_memcpy_h2d(cyb_ctx_t ctx, void* dst, const void* src, size_t bytes)
{
    CYB_REQUIRE(ctx, "cyb_memcpy_h2d: ctx is NULL");
    if (bytes) {
        // pageable source: hipMemcpyAsync stages it, so the host buffer may be reused on return
        CYB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        CYB_HIP(hipStreamSynchronize(ctx->stream));
    }
    return CYB_OK;
}

int cyb_memcpy_d2h(cyb_ctx_t ctx, void* dst, const void* src, size_t bytes)
{
    CYB_REQUIRE(ctx, "cyb_memcpy_d2h: ctx is NULL");
    return ctx->d2h(dst, src, bytes);
}

int cyb_memcpy_d2d(cyb_ctx_t ctx, void* dst, const void* src, size_t bytes)
{
    CYB_REQUIRE(ctx, "cyb_memcpy_d2d: ctx is NULL");
    if (bytes) CYB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return CYB_OK;
}

int cyb_memset(cyb_ctx_t ctx, void* dst, int byte, size_t bytes)
{
    CYB_REQUIRE(ctx, "cyb_memset: ctx is NULL");
    if (bytes) CYB_HIP(hipMemsetAsync(dst, byte, bytes, ctx->stream));
    return CYB_OK;
}

struct cyb_event_s {
    hipEvent_t ev;
};

int cyb_event_create(cyb_event_t* out)
{
    CYB_REQUIRE(out, "cyb_event_create: out is NULL");
    cyb_event_s* e = new cyb_event_s();
    hipError_t err = hipEventCreate(&e->ev);
    if (err != hipSuccess) {
        delete e;
        cyb::set_error("hipEventCreate failed: %s", hipGetErrorString(err));
        return CYB_ERR_HIP;
    }
    *out = e;
    return CYB_OK;
}

int cyb_event_destroy(cyb_event_t ev)
{
    if (ev) {
        (void)hipEventDestroy(ev->ev);
        delete ev;
    }
    return CYB_OK;
}

int cyb_event_record(cyb_ctx_t ctx, cyb_event_t ev)
{
    CYB_REQUIRE(ctx && ev, "cyb_event_record: NULL argument");
    CYB_HIP(hipEventRecord(ev->ev, ctx->stream));
    return CYB_OK;
}

int cyb_ctx_time_next_gemm(cyb_ctx_t ctx, cyb_event_t start, cyb_event_t stop)
{
    CYB_REQUIRE(ctx, "cyb_ctx_time_next_gemm: ctx is NULL");
    ctx->time_start = start ? start->ev : nullptr;
    ctx->time_stop = stop ? stop->ev : nullptr;
    return CYB_OK;
}

int cyb_event_elapsed_ms(cyb_event_t start, cyb_event_t stop, float* ms)
{
    CYB_REQUIRE(start && stop && ms, "cyb_event_elapsed_ms: NULL argument");
    CYB_HIP(hipEventSynchronize(stop->ev));
    CYB_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return CYB_OK;
}

} // extern "C"
