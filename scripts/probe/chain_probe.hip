// Diagnostic (not part of the product): a dependent chain  D0 -> P0 -> D1 -> P1 -> D2  of one-workgroup "diagonal" kernels (busy for
// ~15 us) and many-workgroup "panel" kernels (busy for ~3 us), run
//   serial   all five in ONE stream: every arrow is a kernel boundary
//   overlap  the D kernels in stream 1, the P kernels in stream 2, every arrow an in-memory flag / counter written and polled with
//            agent-scope atomics (no fences): a kernel is launched -- and its launch cost paid -- while its predecessor still runs
// Prints the time per chain for both.  The payload handed over is a 96 x 96 tile (D -> P) and 21 blocks of 256 doubles (P -> D).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void busy(long cycles) {
    const long t0 = (long)__builtin_amdgcn_s_memtime();
    while ((long)__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void wait_eq(const int *f, int want) {
    if (threadIdx.x == 0) while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
}

// D kernel: (flagged) wait for `need` arrivals on cnt_in, read the blocks, work, write the tile, raise flag_out = epoch
__global__ void __launch_bounds__(1024) k_diag(int flagged, const int *cnt_in, int need, const double *blocks, double *tile, int *flag_out, int epoch, long work,
                                                double *sink) {
    double acc = 0.0;
    if (flagged && need) wait_eq(cnt_in, need);
    if (need)
        for (int i = threadIdx.x; i < 21 * 256; i += 1024) acc += flagged ? __hip_atomic_load(blocks + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : blocks[i];
    busy(work);
    for (int i = threadIdx.x; i < 96 * 96; i += 1024) {
        if (flagged) __hip_atomic_store(tile + i, acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else tile[i] = acc + i;
    }
    if (flagged) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag_out, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 12345.678) sink[0] = acc;
}
// P kernel: (flagged) wait for flag_in == epoch, read the tile, work, write one block, count in
__global__ void __launch_bounds__(128) k_panel(int flagged, const int *flag_in, int epoch, const double *tile, double *blocks, int *cnt_out, long work, double *sink) {
    double acc = 0.0;
    if (flagged) wait_eq(flag_in, epoch);
    for (int i = threadIdx.x; i < 96 * 96; i += 128 * 4) acc += flagged ? __hip_atomic_load(tile + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tile[i];
    busy(work);
    if (blockIdx.x < 21) {
        for (int i = threadIdx.x; i < 256; i += 128) {
            double *p = blocks + blockIdx.x * 256 + i;
            if (flagged) __hip_atomic_store(p, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = acc;
        }
    }
    if (flagged) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x < 21) __hip_atomic_fetch_add(cnt_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main() {
    double *tile, *blocks, *sink; int *sync;
    hipMalloc(&tile, 3 * 96 * 96 * 8); hipMalloc(&blocks, 3 * 21 * 256 * 8); hipMalloc(&sink, 64); hipMalloc(&sync, 4096);
    hipMemset(sync, 0, 4096); hipMemset(blocks, 0, 3 * 21 * 256 * 8);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const long wd = 15 * 2400, wp = 3 * 2400;   // ~15 us and ~3 us at 2.4 GHz
    const int reps = 300;
    int *flag = sync, *cnt = sync + 64;          // flag[s], cnt[s]: monotonic (epoch-valued flags, cumulative counters)
    for (int mode = 0; mode < 2; mode++) {
        for (int pass = 0; pass < 2; pass++) {   // pass 0 = warm-up
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, s1);
            for (int r = 0; r < reps; r++) {
                const int epoch = (mode * 2 + pass) * reps + r + 1;
                for (int s = 0; s < 3; s++) {
                    hipStream_t sp = mode ? s2 : s1;
                    hipLaunchKernelGGL(k_diag, dim3(1), dim3(1024), 0, s1, mode, cnt + 16 * (s ? s - 1 : 0), s ? 21 * (mode ? (pass * reps + r + 1) : 0) : 0, blocks + (s ? s - 1 : 0) * 21 * 256,
                                       tile + s * 96 * 96, flag + 16 * s, epoch, wd, sink);
                    if (s < 2) hipLaunchKernelGGL(k_panel, dim3(s == 0 ? 90 : 27), dim3(128), 0, sp, mode, flag + 16 * s, epoch, tile + s * 96 * 96, blocks + s * 21 * 256, cnt + 16 * s, wp, sink);
                }
            }
            hipEventRecord(e1, s1);
            hipEventSynchronize(e1);
            hipStreamSynchronize(s2);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (pass) printf("%s: %.2f us per chain of 3 D (15 us) + 2 P (3 us) kernels [pure work 51 us]\n", mode ? "overlap (2 streams, flags)" : "serial (1 stream)", ms * 1e3 / reps);
        }
    }
    return 0;
}
