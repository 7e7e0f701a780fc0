// Page-locked staging for every host <-> device transfer of libaar: see hostcopy.h.
#include "hostcopy.h"

#include <algorithm>
#include <cstring>

namespace aar {

namespace {

constexpr size_t STAGE_BYTES = (size_t)8 << 20;   // two halves of 4 MiB: the DMA of one chunk overlaps the memcpy of the next

// One staging buffer per host thread (the in-process rank groups of the tests drive one problem per thread; a mutex-guarded shared buffer
// would serialise their copies for nothing), freed when the thread ends.  The buffer is portable page-locked memory (any device may DMA
// through it); the two events belong to ONE device -- an event may only be recorded on a stream of the device it was created on -- so they
// are made again when the thread's current device is another one than last time (a thread that drives problems on devices 0 and 1 in turn).
struct Stage {
    char *buf = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int dev = -1;
    bool gone = false;   // the thread's destructors have run (a process-exit handler still copying: see d2h)
    void drop_events() {
        for (int i = 0; i < 2; i++) { if (ev[i]) (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
        dev = -1;
    }
    ~Stage() {
        drop_events();
        if (buf) (void)hipHostFree(buf);
        buf = nullptr;
        gone = true;
    }
    hipError_t ensure() {
        hipError_t e;
        int cur = 0;
        if ((e = hipGetDevice(&cur)) != hipSuccess) return e;
        if (!buf && (e = hipHostMalloc((void **)&buf, STAGE_BYTES, hipHostMallocPortable)) != hipSuccess) { buf = nullptr; return e; }
        if (dev != cur || !ev[0] || !ev[1]) {
            drop_events();
            for (int i = 0; i < 2; i++)
                if ((e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)) != hipSuccess) { ev[i] = nullptr; drop_events(); return e; }
            dev = cur;
        }
        return hipSuccess;
    }
};
thread_local Stage t_stage;

int fail(hipError_t e, const char *name, const char **what) {
    if (what) *what = name;
    return (int)e;
}

}  // namespace

int h2d(void *dst, const void *src, size_t bytes, hipStream_t st, const char **what) {
    if (bytes == 0) return 0;
    hipError_t e = t_stage.ensure();
    if (e != hipSuccess) return fail(e, "hipHostMalloc(staging)", what);
    const size_t half = STAGE_BYTES / 2;
    bool used[2] = {false, false};
    int h = 0;
    for (size_t off = 0; off < bytes; off += half, h ^= 1) {
        const size_t n = std::min(half, bytes - off);
        char *stg = t_stage.buf + (size_t)h * half;
        if (used[h] && (e = hipEventSynchronize(t_stage.ev[h])) != hipSuccess) return fail(e, "hipEventSynchronize(staging)", what);   // this half's previous chunk has left
        memcpy(stg, (const char *)src + off, n);
        if ((e = hipMemcpyAsync((char *)dst + off, stg, n, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "hipMemcpyAsync(host to device)", what);
        if ((e = hipEventRecord(t_stage.ev[h], st)) != hipSuccess) return fail(e, "hipEventRecord(staging)", what);
        used[h] = true;
    }
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "hipStreamSynchronize(upload)", what);
    return 0;
}

int d2h(void *dst, const void *src, size_t bytes, hipStream_t st, const char **what) {
    if (bytes == 0) return 0;
    if (t_stage.gone) {   // after the thread's destructors (an atexit handler of a diagnostic build): a page-locked landing zone of its own, no staging state
        void *tmp = nullptr;
        hipError_t e2 = hipHostMalloc(&tmp, bytes, hipHostMallocPortable);
        if (e2 != hipSuccess) return fail(e2, "hipHostMalloc(exit-time copy)", what);
        e2 = hipMemcpy(tmp, src, bytes, hipMemcpyDeviceToHost);
        if (e2 == hipSuccess) memcpy(dst, tmp, bytes);
        (void)hipHostFree(tmp);
        return e2 == hipSuccess ? 0 : fail(e2, "hipMemcpy(exit-time copy)", what);
    }
    hipError_t e = t_stage.ensure();
    if (e != hipSuccess) return fail(e, "hipHostMalloc(staging)", what);
    const size_t half = STAGE_BYTES / 2;
    // chunk k travels into half k % 2 while chunk k - 1 is copied out of the other half
    size_t prev_off = 0, prev_n = 0;
    int h = 0;
    for (size_t off = 0; off < bytes; off += half, h ^= 1) {
        const size_t n = std::min(half, bytes - off);
        char *stg = t_stage.buf + (size_t)h * half;
        if ((e = hipMemcpyAsync(stg, (const char *)src + off, n, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e, "hipMemcpyAsync(device to host)", what);
        if ((e = hipEventRecord(t_stage.ev[h], st)) != hipSuccess) return fail(e, "hipEventRecord(staging)", what);
        if (prev_n) {
            if ((e = hipEventSynchronize(t_stage.ev[h ^ 1])) != hipSuccess) return fail(e, "hipEventSynchronize(staging)", what);
            memcpy((char *)dst + prev_off, t_stage.buf + (size_t)(h ^ 1) * half, prev_n);
        }
        prev_off = off; prev_n = n;
    }
    if ((e = hipEventSynchronize(t_stage.ev[h ^ 1])) != hipSuccess) return fail(e, "hipEventSynchronize(staging)", what);
    memcpy((char *)dst + prev_off, t_stage.buf + (size_t)(h ^ 1) * half, prev_n);
    return 0;
}

int PinnedBuf::reserve(size_t count) {
    if (count <= n && p) return 0;
    release();
    const size_t want = std::max<size_t>(count, 8);
    hipError_t e = hipHostMalloc((void **)&p, want * sizeof(double), hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; n = 0; return (int)e; }
    n = want;
    memset(p, 0, n * sizeof(double));
    return 0;
}

void PinnedBuf::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = 0;
}

}  // namespace aar
