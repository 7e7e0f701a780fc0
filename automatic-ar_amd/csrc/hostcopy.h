// ONE copy discipline for host memory (every translation unit of libaar goes through these two functions for every transfer between
// caller- or heap-owned host memory and the device):
//
//   h2d(dst_device, src_host, bytes, stream)      host -> device
//   d2h(dst_host, src_device, bytes, stream)      device -> host
//
// Both move the bytes through PAGE-LOCKED staging memory the library owns (hipHostMalloc, one buffer per host thread, kept for the thread's
// life) in chunks, and both are synchronous towards the host buffer: when they return the source may be reused / the destination holds the
// data.  The HIP runtime is never handed a pageable pointer, so it never pins heap pages on the fly.  Why that matters (rounds 3-4,
// profiles/r04_attempts.txt): for an upload from pageable memory the runtime pins the source pages READ-ONLY for the GPU and keeps the pinning
// cached; once malloc has handed the same address range out again, a later device-to-host copy INTO it is a GPU write to a page the GPU may
// only read -- "write access to a read-only page", raised on the HSA event thread (SIGABRT), about once per six runs of the GPU test suite.
//
// Ordering: the copies are issued on `stream` and waited for on it, so they are ordered behind the work already queued there and
// the caller's next launch on that stream sees an upload complete.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace aar {

// 0 on success, otherwise the hipError_t of the failing call (as int); *what (optional) names it
int h2d(void *dst_device, const void *src_host, size_t bytes, hipStream_t stream, const char **what = nullptr);
int d2h(void *dst_host, const void *src_device, size_t bytes, hipStream_t stream, const char **what = nullptr);

// a page-locked host buffer owned by its holder (pose staging of a problem, landing zones): allocated once, freed with the holder
struct PinnedBuf {
    double *p = nullptr;
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    int reserve(size_t count);          // at least `count` doubles (contents are NOT kept when it grows); 0 or a hipError_t
    void release();
    double *data() { return p; }
    size_t size() const { return n; }
};

}  // namespace aar
