// Diagnostic (not part of the product): what does it cost to hand a small payload from one workgroup to another?
//   mode 0  release / acquire fences at agent scope around plain stores / loads (what k_ldl_backsolve's chain does)
//   mode 1  no fences: the payload itself travels as relaxed agent-scope atomic stores / loads (sc1: past the XCD's L2),
//           the writer drains its stores (s_waitcnt) before it raises the flag
//   chain   the same hand-over as a kernel boundary: a chain of dependent one-workgroup kernels in one stream
// Two workgroups play ping-pong; `partner` picks the second one's index (workgroup i runs on XCD i mod 8: partner 1 = the next
// XCD, partner 8 = the same XCD, another CU); everybody else leaves at once.  Payload: n doubles, checked on arrival.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void wait_flag(const int *f, int want, int mode) {
    if (threadIdx.x == 0) {
        if (mode == 0) while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
        else while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (mode == 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
__device__ __forceinline__ void raise_flag(int *f, int v, int mode) {
    if (mode == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __builtin_amdgcn_s_waitcnt(0);   // this thread's stores have been acknowledged at the coherence point
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void put(double *p, double v, int mode) {
    if (mode == 0) *p = v; else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double get(const double *p, int mode) {
    return mode == 0 ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(1024) k_pingpong(int mode, int rounds, int n, double *bufA, double *bufB, int *flags, unsigned long long *out,
                                                   int partner) {
    const int me = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == partner ? 1 : -1);
    if (me < 0) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    int bad = 0;
    unsigned long long t0 = 0;
    for (int r = 1; r <= rounds; r++) {
        if (r == 11 && tid == 0) t0 = wall_clock64();
        if (me == 0) {
            for (int i = tid; i < n; i += nt) put(bufA + i, (double)(r * 4096 + i), mode);
            raise_flag(flags, r, mode);
            wait_flag(flags + 32, r, mode);
            for (int i = tid; i < n; i += nt) bad += get(bufB + i, mode) != (double)(r * 4096 + i + 1);
        } else {
            wait_flag(flags, r, mode);
            for (int i = tid; i < n; i += nt) { const double v = get(bufA + i, mode); bad += v != (double)(r * 4096 + i); put(bufB + i, v + 1.0, mode); }
            raise_flag(flags + 32, r, mode);
        }
    }
    if (tid == 0 && me == 0) out[0] = wall_clock64() - t0;
    atomicAdd((unsigned int *)(out + 1), (unsigned)bad);
}

__global__ void __launch_bounds__(1024) k_chain(int n, const double *src, double *dst) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i] + 1.0;
}

int main(int argc, char **argv) {
    const int rounds = 2010;
    double *a, *b; int *fl; unsigned long long *out, h[2];
    hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMalloc(&fl, 1024); hipMalloc(&out, 64);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
    printf("wall clock %d kHz\n", rate);
    for (int threads : {256, 1024})
        for (int n : {96, 1024, 9216})
            for (int partner : {1, 8, 4})
                for (int mode : {0, 1}) {
                    hipMemset(fl, 0, 1024); hipMemset(out, 0, 64);
                    hipLaunchKernelGGL(k_pingpong, dim3(partner + 1), dim3(threads), 0, 0, mode, rounds, n, a, b, fl, out, partner);
                    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
                    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
                    printf("threads %4d payload %6d B partner wg %d mode %d (%s): %.2f us per hop, %llu payload errors\n", threads, n * 8, partner, mode,
                           mode ? "sc1 atomics, no fence" : "release/acquire fences", (double)h[0] / rate * 1e3 / (2.0 * (rounds - 10)), h[1]);
                }
    for (int threads : {256, 1024})
        for (int n : {96, 1024, 9216}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 100; i++) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, 0, n, a, b); hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, 0, n, b, a); }
            hipEventRecord(e0);
            for (int i = 0; i < 1000; i++) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, 0, n, a, b); hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, 0, n, b, a); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("threads %4d payload %6d B kernel chain: %.2f us per hop\n", threads, n * 8, ms * 1e3 / 2000);
        }
    return 0;
}
